"""The N>1 path on CPU: world_size-2 gloo run of the sharded HipVS (row-sharded corpus, replicated queries,
all-gather of the per-shard candidate keys, merge) with the oracle-backed test double standing in for the GPU."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import synth

HERE = os.path.dirname(os.path.abspath(__file__))


def _worker(rank, world, port, tmp, out_q):
    sys.path.insert(0, os.path.dirname(HERE))
    sys.path.insert(0, HERE)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from lotus_amd import HipVS
        from oracle_backend import OracleBackend

        xb = synth.corpus(1001, 32, seed=21)
        xq, _ = synth.queries(xb, 40, seed=2)
        vs = HipVS(backend=OracleBackend(), shard=True, storage="fp16")
        vs.index(None, xb, os.path.join(tmp, "idx"))
        ent = vs._resident[vs.index_dir]
        full = vs(xq, 7)
        ids = np.random.default_rng(5).choice(1001, 300, replace=False).tolist()
        sub = vs(xq, 7, ids=ids)
        big = vs(xq[:3], 64, ids=ids[:40])  # K beyond the subset, merged across shards
        out_q.put((rank, ent.lo, ent.hi, full.indices, full.distances, sub.indices, sub.distances, big.indices))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_sharded_search_equals_single_process(tmp_path):
    import oracle
    from oracle_backend import _emulate_storage

    world = 2
    port = 29500 + (os.getpid() % 2000)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, str(tmp_path), q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=180) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    xb = synth.corpus(1001, 32, seed=21)
    xq, _ = synth.queries(xb, 40, seed=2)
    xb16, xq16 = _emulate_storage(xb, 0), _emulate_storage(xq, 0)
    D, I = oracle.flat_search(xb16, xq16, 7)
    ids = np.random.default_rng(5).choice(1001, 300, replace=False).tolist()
    Ds, Is = oracle.flat_search(xb16, xq16, 7, ids=ids)
    assert (res[0][1], res[0][2], res[1][1], res[1][2]) == (0, 501, 501, 1001)  # contiguous row shards
    for r in res:
        assert np.array_equal(r[3], I) and np.allclose(r[4], D, atol=1e-6)  # every rank holds the merged result
        assert np.array_equal(r[5], Is) and np.allclose(r[6], Ds, atol=1e-6)
        assert (r[7][:, 40:] == -1).all() and sorted(r[7][0, :40].tolist()) == sorted(ids[:40])


def _worker_km(rank, world, port, out_q):
    sys.path.insert(0, os.path.dirname(HERE))
    sys.path.insert(0, HERE)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from lotus_amd.cluster import kmeans
        from lotus_amd.dedup import threshold_pairs
        from oracle_backend import OracleBackend, _emulate_storage

        rng = np.random.default_rng(3)
        c = rng.standard_normal((5, 12)).astype(np.float32) * 4
        x = _emulate_storage((c[rng.integers(0, 5, 700)] + 0.3 * rng.standard_normal((700, 12))).astype(np.float32), 1)
        r = kmeans(x, 5, niter=5, max_points_per_centroid=64, backend=OracleBackend(), shard=True)
        xd = synth.corpus(300, 16, seed=8)
        xd[150:200] = xd[:50] + 0.02 * synth.corpus(50, 16, seed=9)
        xd /= np.linalg.norm(xd, axis=1, keepdims=True)
        be = OracleBackend()
        i, j, s_ = threshold_pairs(be, be.pack(xd, 1), 0.97, shard=True)
        out_q.put((rank, r.centroids, r.assign, r.obj, i, j))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_sharded_kmeans_and_dedup_equal_single_process():
    import oracle
    from lotus_amd.cluster import kmeans
    from lotus_amd.dedup import threshold_pairs
    from oracle_backend import OracleBackend, _emulate_storage

    world = 2
    port = 31500 + (os.getpid() % 2000)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_km, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=180) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    rng = np.random.default_rng(3)
    c = rng.standard_normal((5, 12)).astype(np.float32) * 4
    x = _emulate_storage((c[rng.integers(0, 5, 700)] + 0.3 * rng.standard_normal((700, 12))).astype(np.float32), 1)
    one = kmeans(x, 5, niter=5, max_points_per_centroid=64, backend=OracleBackend())
    xd = synth.corpus(300, 16, seed=8)
    xd[150:200] = xd[:50] + 0.02 * synth.corpus(50, 16, seed=9)
    xd /= np.linalg.norm(xd, axis=1, keepdims=True)
    be = OracleBackend()
    i1, j1, _ = threshold_pairs(be, be.pack(xd, 1), 0.97)
    assert len(i1) >= 50
    for r in res:
        # the all-reduced sums are added in a different order than the single-process in-row-order sum: tiny drift
        assert np.allclose(r[1], one.centroids, atol=1e-5) and (r[2] == one.assign).mean() >= 0.999
        assert np.allclose(r[3], one.obj, rtol=1e-5)
        assert np.array_equal(r[4], i1) and np.array_equal(r[5], j1)
