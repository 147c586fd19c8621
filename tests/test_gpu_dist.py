"""The N>1 path on the REAL HIP backend: two processes share cuda:0, rendezvous over gloo, and
run every sharded operator - row-sharded search with id offsets, device-side merge of the gathered candidate lists
(short and long lists), sharded K = N ranking / score rows, k-means on a row-sharded index (device sums all-reduced,
per-shard final assignment), tile-dealt dedup - against the single-process CPU oracle.  On an 8-GPU node the only
difference is the transport inside ``lotus_amd/_dist.py`` (RCCL on device tensors instead of host staging)."""
import pytest

import dist_cases

pytestmark = pytest.mark.gpu


def test_sharded_operators_on_hip_backend_two_ranks_one_gpu(tmp_path):
    res = dist_cases.run(tmp_path, "hip")
    dist_cases.check(res, exact=False)


def test_2d_split_on_the_hip_backend(tmp_path):
    """HipVS(shard=(2, 2)): four processes (2 query groups x 2 corpus shards) on cuda:0 - merge inside a corpus group,
    concatenation across the query groups, k-means and score rows through the corpus sub-group."""
    res = dist_cases.run_2d(tmp_path, "hip")
    dist_cases.check_2d(res, exact=False)


def test_shard_scale_agreement_and_query_validation_on_the_hip_backend(tmp_path):
    """fp32 rows with per-shard magnitudes: one agreed pack exponent (re-agreed after an outlier row), and the validation
    verdict of one rank's query slice taken by both ranks - on the real packers / flag words."""
    res = dist_cases.run_scale(tmp_path, "hip")
    dist_cases.check_scale(res, exact=False)


def _rccl_worker(port, out_q):
    import os

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    import torch.distributed as dist

    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        from lotus_amd import _dist

        dev = torch.device("cuda", 0)
        keys = torch.arange(12, dtype=torch.int64, device=dev).reshape(4, 3) * (1 << 40)
        g = _dist.all_gather_rows(keys)  # device tensor straight into RCCL (no host staging under nccl)
        sums = torch.ones((5, 7), dtype=torch.float32, device=dev)
        counts = torch.full((5,), 2.0, dtype=torch.float32, device=dev)
        obj = torch.tensor([3.5], dtype=torch.float64, device=dev)
        _dist.all_reduce_sum_([sums, counts, obj])
        _dist.barrier()
        staged = _dist._staging_device(keys, None)
        out_q.put((tuple(g.shape), bool(torch.equal(g[0], keys)), g.is_cuda, float(sums.sum()), float(counts.sum()), float(obj.item()),
                   staged is None))
    finally:
        dist.destroy_process_group()


def test_rccl_transport_branch_runs_on_device_tensors():
    """The `nccl` (= RCCL) branch of lotus_amd/_dist.py - device tensors handed to the collectives as they are - executed on
    the one GPU of this box (world size 1: RCCL has no second device here; the 8-GPU run is the driver's).  Covers the API
    shapes the path uses: all_gather_into_tensor of int64 keys, all-reduce of float32 sums / counts and the float64 objective."""
    import os

    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_rccl_worker, args=(30500 + os.getpid() % 2000, q))
    p.start()
    shape, same, on_gpu, s, c, o, direct = q.get(timeout=300)
    p.join(60)
    assert p.exitcode == 0
    assert shape == (1, 4, 3) and same and on_gpu and direct
    assert s == 35.0 and c == 10.0 and o == 3.5


def test_bench_self_launches_its_ranks_and_prints_one_line():
    """`python bench.py --gpus 2 ...` with no launcher around it (the form of the driver's N = 1 command): bench.py starts its
    own two ranks under torch.distributed.run, rank 0 prints the one JSON line.  Rehearsal mode (both ranks on cuda:0 over
    gloo) because this box has one GPU - the code path is the N > 1 path of the real run: shared inputs through /dev/shm,
    pooled sample thresholds, seeded shard search, key all-gather, merge, max-over-ranks timing, oracle check on rank 0."""
    import json
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, LOTUS_BENCH_REHEARSAL="1")
    for var in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR"):
        env.pop(var, None)
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--queries", "4096", "--corpus", "262144", "--check-sample", "128"],
                       env=env, cwd=root, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 2 and out["unit"] == "queries/s"
    assert out["roofline"]["bound"] == "mfma" and out["roofline"]["frac"] > 0
    assert out["recall_at_k"] == 1.0 and out["id_mismatches_outside_near_ties"] == 0
    assert out["planted_neighbour_at_rank1"] > 0.99
    assert "REHEARSAL" in out["data"]
    assert not [f for f in os.listdir("/dev/shm") if f.startswith("lotus_bench_")]
