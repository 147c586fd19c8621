"""The N>1 path on CPU: world_size-2 gloo run of every sharded operator (``tests/dist_cases.py``) with the
oracle-backed test double standing in for the GPU.  The same scenarios run on the real HIP backend in
``test_gpu_dist.py``."""
import dist_cases


def test_sharded_operators_equal_single_process(tmp_path):
    res = dist_cases.run(tmp_path, "oracle")
    dist_cases.check(res, exact=True)


def test_2d_split_on_four_ranks(tmp_path):
    """2 query groups x 2 corpus shards (HipVS(shard=(2, 2))): merge inside a corpus group, concatenation across the
    query groups, sub-groups for the k-means / score-row collectives."""
    res = dist_cases.run_2d(tmp_path, "oracle")
    dist_cases.check_2d(res, exact=True)


def test_shard_scale_agreement_and_query_validation_consensus(tmp_path):
    """fp32 rows whose magnitudes differ per shard get ONE pack exponent (re-agreed when a late row exceeds the sampled
    head's headroom); under the query split a validation verdict reached on one rank's slice (retry with another
    exponent, or raise) is taken by every rank."""
    res = dist_cases.run_scale(tmp_path, "oracle")
    dist_cases.check_scale(res, exact=True)
