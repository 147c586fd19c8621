"""The N>1 path on CPU: world_size-2 gloo run of every sharded operator (``tests/dist_cases.py``) with the
oracle-backed test double standing in for the GPU.  The same scenarios run on the real HIP backend in
``test_gpu_dist.py``."""
import dist_cases


def test_sharded_operators_equal_single_process(tmp_path):
    res = dist_cases.run(tmp_path, "oracle")
    dist_cases.check(res, exact=True)
