"""World-size-2 scenarios of the sharded path, shared by the CPU run (gloo + the oracle-backed test double,
``test_dist_gloo.py``) and the GPU run (gloo rendezvous, BOTH ranks on cuda:0 with the real ``HipBackend``,
``test_gpu_dist.py``).  The code under test is the same in both: shard offsets, per-shard search, all-gather of the
candidate keys, device-side merge, sharded K = N ranking and score rows, sharded k-means (all-reduce of sums / counts,
per-shard final assignment), tile-dealt dedup.  Only the collective's transport differs from an 8-GPU node (host
staging instead of RCCL, ``lotus_amd/_dist.py``)."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))

NB, D, NQ = 5001, 64, 300
KM_N, KM_D, KM_K = 3000, 32, 8


def km_data():
    rng = np.random.default_rng(3)
    c = rng.standard_normal((KM_K, KM_D)).astype(np.float32) * 4
    x = (c[rng.integers(0, KM_K, KM_N)] + 0.3 * rng.standard_normal((KM_N, KM_D))).astype(np.float32)
    return x.astype(np.float16)  # fp16 storage: the device image holds exactly these values


def dedup_data():
    import synth

    xd = synth.corpus(1500, 32, seed=8)
    xd[700:900] = xd[:200] + 0.02 * synth.corpus(200, 32, seed=9)
    xd /= np.linalg.norm(xd, axis=1, keepdims=True)
    return xd.astype(np.float16)


def search_data():
    import synth

    xb = synth.corpus(NB, D, seed=21)
    xq, _ = synth.queries(xb, NQ, seed=2)
    return xb.astype(np.float16), xq.astype(np.float16)


SEED_NB, SEED_D, SEED_NQ, SEED_K = 90_001, 32, 2100, 10


def seeded_data():
    """A row-sharded join large enough for the pooled sample thresholds (lvs_flat_search_seed_scores + one all-gather +
    lvs_flat_search_keys_seeded): >= 2048 queries, shards of >= 16 x 256 x k rows."""
    import synth

    xb = synth.corpus(SEED_NB, SEED_D, seed=31)
    xq, _ = synth.queries(xb, SEED_NQ, seed=32)
    return xb.astype(np.float16), xq.astype(np.float16)


def subset_ids():
    return np.random.default_rng(5).choice(NB, 1500, replace=False).tolist()


def subset_big():
    return np.random.default_rng(6).choice(NB, 2600, replace=False).tolist()


def worker(rank, world, port, tmp, out_q, backend_kind):
    sys.path.insert(0, os.path.dirname(HERE))
    sys.path.insert(0, HERE)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    import torch.distributed as dist

    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from lotus_amd import HipVS
        from lotus_amd.cluster import kmeans
        from lotus_amd.dedup import threshold_pairs

        if backend_kind == "hip":
            from lotus_amd.backend import HipBackend

            torch.cuda.set_device(0)  # both ranks share the one GPU of the test box
            be = HipBackend("cuda:0")
        else:
            from oracle_backend import OracleBackend

            be = OracleBackend()
        res = {"rank": rank}
        xb, xq = search_data()
        vs = HipVS(backend=be, shard=True)
        vs.index(None, xb, os.path.join(tmp, "idx"))
        ent = vs._resident[vs.index_dir]
        res["bounds"] = (ent.lo, ent.hi, ent.packed.n)
        ids = subset_ids()
        for name, out in (("full", vs(xq, 7)), ("sub", vs(xq, 7, ids=ids)), ("pad", vs(xq[:3], 64, ids=ids[:40])),
                          ("k1000", vs(xq[:40], 1000)), ("rank_all", vs(xq[:5], 3000)),
                          ("k1500_sub", vs(xq[:4], 1500, ids=ids)), ("rank_sub", vs(xq[:4], 2600, ids=subset_big()))):
            res[name] = (np.asarray(out.distances), np.asarray(out.indices))
        # the other split: corpus replicated, every rank searches its slice of the queries (299 = uneven slices)
        vq = HipVS(backend=be, shard="queries")
        vq.load_index(os.path.join(tmp, "idx"))
        entq = vq._resident[vq.index_dir]
        res["q_bounds"] = (entq.lo, entq.hi, entq.packed.n)
        for name, out in (("q_full", vq(xq[:299], 7)), ("q_sub", vq(xq[:299], 7, ids=ids)), ("q_one", vq(xq[:1], 5))):
            res[name] = (np.asarray(out.distances), np.asarray(out.indices))
        # the same two splits written as (query groups, corpus shards) pairs, and the planner's choice for 2 ranks
        for name, shard in (("t_1x2", (1, 2)), ("t_2x1", (2, 1)), ("t_auto", "auto")):
            vt = HipVS(backend=be, shard=shard)
            vt.load_index(os.path.join(tmp, "idx"))
            out = vt(xq[:299], 7)
            entt = vt._resident[vt.index_dir]
            res[name] = (np.asarray(out.distances), np.asarray(out.indices), (entt.lo, entt.hi))
        # index() under the query split: every rank holds the whole corpus, ONE rank writes the directory
        import lotus_amd.store as store_mod

        writes, orig_write = [], store_mod.write_dir
        store_mod.write_dir = lambda *a, **k: (writes.append(a[0]), orig_write(*a, **k))[1]
        vq2 = HipVS(backend=be, shard="queries")
        vq2.index(None, xb[:500], os.path.join(tmp, "qidx"))
        store_mod.write_dir = orig_write
        res["q_index"] = (len(writes), vq2._resident[vq2.index_dir].sig, np.asarray(vq2(xq[:9], 3).indices))
        res["scores"] = vs.scores(xq[:6])
        res["scores_sub"] = vs.scores(xq[:6], ids=ids[:77])
        # a join big enough for the pooled sample thresholds: every shard scores a sample, one all-gather, seeded search
        xsb, xsq = seeded_data()
        calls = []
        orig_seed = be.seed_scores
        be.seed_scores = lambda *a, **k: (calls.append(int(a[3])), orig_seed(*a, **k))[1]
        vseed = HipVS(backend=be, shard=True)
        vseed.index(None, xsb, os.path.join(tmp, "seeded"), persist=False)
        out = vseed(xsq, SEED_K)
        out_small = vseed(xsq[:100], SEED_K)  # below the exchange's query floor: plain sharded search
        be.seed_scores = orig_seed
        res["seeded"] = (np.asarray(out.distances), np.asarray(out.indices), list(calls))
        res["seeded_small"] = (np.asarray(out_small.distances), np.asarray(out_small.indices))
        if backend_kind == "hip":
            # the same join with both exchanges issued from INSIDE the C ABI (lvs_search_sharded; this process group's all-gather is
            # the callback it is handed): same kernels, same order - the same arrays
            vabi = HipVS(backend=be, shard=True, abi_exchange=True)
            vabi.index(None, xsb, os.path.join(tmp, "seeded_abi"), persist=False)
            oa, ob = vabi(xsq, SEED_K), vabi(xb_small_q := xsq[:100], SEED_K)
            res["seeded_abi"] = (np.asarray(oa.distances), np.asarray(oa.indices), np.asarray(ob.distances), np.asarray(ob.indices))
        # k-means on the row-sharded index: all rows, then a subset of rows
        xk = km_data()
        vk = HipVS(backend=be, shard=True)
        vk.index(None, xk, os.path.join(tmp, "km"))
        r = vk.kmeans(None, KM_K, niter=6, return_result=True, max_points_per_centroid=128)
        res["km"] = (r.centroids, r.assign, r.obj, r.train_ids)
        kid = np.random.default_rng(11).choice(KM_N, 1200, replace=False)
        r2 = vk.kmeans(None, 4, niter=4, ids=kid.tolist(), return_result=True)
        res["km_sub"] = (r2.centroids, r2.assign, r2.obj)
        # replicated rows, training rows / assignment dealt to the ranks
        r3 = kmeans(xk, KM_K, niter=3, backend=be, shard=True, max_points_per_centroid=64)
        res["km_rep"] = (r3.centroids, r3.assign, r3.obj)
        # dedup: rows replicated, 256-query tiles dealt round-robin
        xd = dedup_data()
        i, j, s_ = threshold_pairs(be, be.pack(xd, 0), 0.97, shard=True)
        res["dedup"] = (i, j, s_)
        out_q.put(res)
        dist.barrier()
    finally:
        dist.destroy_process_group()


def run(tmp, backend_kind, world=2):
    import torch.multiprocessing as mp

    port = 29500 + (os.getpid() % 2000) + (7 if backend_kind == "hip" else 0)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=worker, args=(r, world, port, str(tmp), q, backend_kind)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=600) for _ in range(world)], key=lambda t: t["rank"])
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    return res


def check(res, exact: bool):
    """Compare every rank's results with the single-process CPU oracle (``exact``: bit-level, for the test double)."""
    import oracle
    import synth
    from lotus_amd.cluster import kmeans
    from lotus_amd.dedup import threshold_pairs
    from oracle_backend import OracleBackend

    xb, xq = search_data()
    xb32, xq32 = xb.astype(np.float32), xq.astype(np.float32)
    ids = subset_ids()
    per = -(-NB // 2)
    assert [r["bounds"] for r in res] == [(0, per, per), (per, NB, NB - per)]  # contiguous row shards, only the shard resident

    def same_topk(got, ref, k):
        D, I = got
        Dr, Ir = ref
        if exact:
            assert np.array_equal(I, Ir) and np.allclose(D, Dr, atol=1e-6)
        else:
            err, hard, recall = synth.compare_topk(Dr, Ir, D, I, atol=1e-5)
            assert err <= 1e-5 and hard == 0 and recall >= 0.9999, (err, hard, recall)

    ref_full = oracle.flat_search(xb32, xq32, 7)
    ref_sub = oracle.flat_search(xb32, xq32, 7, ids=ids)
    ref_1000 = oracle.flat_search(xb32, xq32[:40], 1000)
    ref_all = oracle.flat_search(xb32, xq32[:5], 3000)
    ref_1500 = oracle.flat_search(xb32, xq32[:4], 1500, ids=ids)
    ref_rsub = oracle.flat_search(xb32, xq32[:4], 2600, ids=subset_big())
    S = xq32[:6] @ xb32.T
    for r in res:
        same_topk(r["full"], ref_full, 7)
        same_topk(r["sub"], ref_sub, 7)
        Ip = r["pad"][1]
        assert (Ip[:, 40:] == -1).all() and sorted(Ip[0, :40].tolist()) == sorted(ids[:40])
        same_topk(r["k1000"], ref_1000, 1000)   # 2 shards x 1000 keys: the long-list merge
        same_topk(r["rank_all"], ref_all, 3000)  # K > LVS_MAX_K on a sharded index: gathered score rows + sort
        same_topk(r["k1500_sub"], ref_1500, 1500)
        same_topk(r["rank_sub"], ref_rsub, 2600)  # ... and on a subset of the rows (ids remapped through the shards)
        assert r["scores"].shape == (6, NB) and np.abs(r["scores"] - S).max() <= 1e-5
        assert np.abs(r["scores_sub"] - S[:, ids[:77]]).max() <= 1e-5
    xsb, xsq = seeded_data()
    ref_seed = oracle.flat_search(xsb.astype(np.float32), xsq.astype(np.float32), SEED_K)
    for r in res:
        same_topk(r["seeded"][:2], ref_seed, SEED_K)
        assert len(r["seeded"][2]) == 1 and r["seeded"][2][0] >= SEED_K  # ONE exchange, only for the big call
        same_topk(r["seeded_small"], (ref_seed[0][:100], ref_seed[1][:100]), SEED_K)
    assert np.array_equal(res[0]["seeded"][1], res[1]["seeded"][1])
    for r in res:
        if "seeded_abi" in r:  # HIP backend only: the exchange inside the C ABI gives the very same arrays
            assert np.array_equal(r["seeded_abi"][0], r["seeded"][0]) and np.array_equal(r["seeded_abi"][1], r["seeded"][1])
            assert np.array_equal(r["seeded_abi"][2], r["seeded_small"][0]) and np.array_equal(r["seeded_abi"][3], r["seeded_small"][1])
    ref_qf = oracle.flat_search(xb32, xq32[:299], 7)
    ref_qs = oracle.flat_search(xb32, xq32[:299], 7, ids=ids)
    ref_q1 = oracle.flat_search(xb32, xq32[:1], 5)
    for r in res:
        assert r["q_bounds"] == (0, NB, NB)  # query split: the whole corpus on every rank
        same_topk(r["q_full"], ref_qf, 7)
        same_topk(r["q_sub"], ref_qs, 7)
        same_topk(r["q_one"], ref_q1, 5)    # fewer queries than ranks: one rank's slice is empty
    for r in res:
        same_topk(r["t_1x2"][:2], ref_qf, 7)
        same_topk(r["t_2x1"][:2], ref_qf, 7)
        same_topk(r["t_auto"][:2], ref_qf, 7)
        assert r["t_2x1"][2] == (0, NB)
    assert [r["t_1x2"][2] for r in res] == [(0, per), (per, NB)]
    ref_qi = oracle.flat_search(xb32[:500], xq32[:9], 3)
    assert [r["q_index"][0] for r in res] == [1, 0]           # only the group's rank 0 wrote the directory
    assert res[0]["q_index"][1] == res[1]["q_index"][1]       # ... and both recorded the finished directory's signature
    for r in res:
        assert np.array_equal(r["q_index"][2], ref_qi[1])
    # every rank holds the same merged answers
    for key in ("full", "sub", "k1000", "rank_all"):
        assert np.array_equal(res[0][key][1], res[1][key][1]) and np.array_equal(res[0][key][0], res[1][key][0])

    # ---- k-means: sharded == single process (up to the summation order of the all-reduced partial sums) ----
    ob = OracleBackend()
    xk = km_data()
    one = kmeans(xk, KM_K, niter=6, backend=ob, max_points_per_centroid=128)
    kid = np.random.default_rng(11).choice(KM_N, 1200, replace=False)
    one_sub = kmeans(xk[kid], 4, niter=4, backend=ob)
    one_rep = kmeans(xk, KM_K, niter=3, backend=ob, max_points_per_centroid=64)
    for r in res:
        c, a, o, tid = r["km"]
        assert np.array_equal(tid, one.train_ids)
        assert np.allclose(c, one.centroids, atol=2e-5) and (a == one.assign).mean() >= 0.999 and len(a) == KM_N
        assert np.allclose(o, one.obj, rtol=1e-5)
        c, a, o = r["km_sub"]
        assert np.allclose(c, one_sub.centroids, atol=2e-5) and (a == one_sub.assign).mean() >= 0.999 and len(a) == 1200
        assert np.allclose(o, one_sub.obj, rtol=1e-5)
        c, a, o = r["km_rep"]
        assert np.allclose(c, one_rep.centroids, atol=2e-5) and (a == one_rep.assign).mean() >= 0.999
    assert np.array_equal(res[0]["km"][1], res[1]["km"][1])

    # ---- dedup pairs ----
    xd = dedup_data()
    i1, j1, s1 = threshold_pairs(ob, ob.pack(xd, 0), 0.97)
    assert len(i1) >= 200
    for r in res:
        i, j, s_ = r["dedup"]
        if exact:
            assert np.array_equal(i, i1) and np.array_equal(j, j1)
        else:  # pairs within 2e-5 of the threshold may differ (summation order)
            got, ref = set(zip(i.tolist(), j.tolist())), set(zip(i1.tolist(), j1.tolist()))
            sd = (xd.astype(np.float32) @ xd.astype(np.float32).T)
            for a, b in got ^ ref:
                assert abs(sd[a, b] - 0.97) <= 2e-5, (a, b, sd[a, b])


# ---- 2-D split: 2 query groups x 2 corpus shards on 4 ranks ------------------------------------------------------------
def worker_2d(rank, world, port, tmp, out_q, backend_kind):
    sys.path.insert(0, os.path.dirname(HERE))
    sys.path.insert(0, HERE)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    import torch.distributed as dist

    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from lotus_amd import HipVS

        if backend_kind == "hip":
            from lotus_amd.backend import HipBackend

            torch.cuda.set_device(0)
            be = HipBackend("cuda:0")
        else:
            from oracle_backend import OracleBackend

            be = OracleBackend()
        xb, xq = search_data()
        vs = HipVS(backend=be, shard=(2, 2))
        vs.index(None, xb, os.path.join(tmp, "idx2d"))
        ent = vs._resident[vs.index_dir]
        res = {"rank": rank, "bounds": (ent.lo, ent.hi, ent.packed.n), "layout": vs._layout()[:4]}
        ids = subset_ids()
        for name, out in (("full", vs(xq[:299], 7)), ("sub", vs(xq[:299], 7, ids=ids)), ("one", vs(xq[:1], 5)),
                          ("k1000", vs(xq[:40], 1000))):
            res[name] = (np.asarray(out.distances), np.asarray(out.indices))
        res["scores"] = vs.scores(xq[:6])
        xk = km_data()
        vk = HipVS(backend=be, shard=(2, 2))
        vk.index(None, xk, os.path.join(tmp, "km2d"))
        r = vk.kmeans(None, KM_K, niter=4, return_result=True, max_points_per_centroid=128)
        res["km"] = (r.centroids, r.assign, r.obj)
        out_q.put(res)
        dist.barrier()
    finally:
        dist.destroy_process_group()


def run_2d(tmp, backend_kind):
    import torch.multiprocessing as mp

    world = 4
    port = 31500 + (os.getpid() % 2000) + (7 if backend_kind == "hip" else 0)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=worker_2d, args=(r, world, port, str(tmp), q, backend_kind)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=600) for _ in range(world)], key=lambda t: t["rank"])
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    return res


def check_2d(res, exact: bool):
    import oracle
    import synth
    from lotus_amd.cluster import kmeans
    from oracle_backend import OracleBackend

    xb, xq = search_data()
    xb32, xq32 = xb.astype(np.float32), xq.astype(np.float32)
    per = -(-NB // 2)
    # rank r: query group r // 2, corpus shard r % 2
    assert [r["layout"] for r in res] == [(0, 2, 0, 2), (0, 2, 1, 2), (1, 2, 0, 2), (1, 2, 1, 2)]
    assert [r["bounds"][:2] for r in res] == [(0, per), (per, NB), (0, per), (per, NB)]

    def same_topk(got, ref):
        D, I = got
        Dr, Ir = ref
        if exact:
            assert np.array_equal(I, Ir) and np.allclose(D, Dr, atol=1e-6)
        else:
            err, hard, recall = synth.compare_topk(Dr, Ir, D, I, atol=1e-5)
            assert err <= 1e-5 and hard == 0 and recall >= 0.9999, (err, hard, recall)

    ids = subset_ids()
    refs = {"full": oracle.flat_search(xb32, xq32[:299], 7), "sub": oracle.flat_search(xb32, xq32[:299], 7, ids=ids),
            "one": oracle.flat_search(xb32, xq32[:1], 5), "k1000": oracle.flat_search(xb32, xq32[:40], 1000)}
    S = xq32[:6] @ xb32.T
    for r in res:
        for name, ref in refs.items():
            same_topk(r[name], ref)  # every rank ends with the complete answer
        assert np.abs(r["scores"] - S).max() <= 1e-5
    one = kmeans(km_data(), KM_K, niter=4, backend=OracleBackend(), max_points_per_centroid=128)
    for r in res:  # k-means runs inside a corpus group (2 shards); the two query groups repeat it
        c, a, o = r["km"]
        assert np.allclose(c, one.centroids, atol=2e-5) and (a == one.assign).mean() >= 0.999 and np.allclose(o, one.obj, rtol=1e-5)


# ---- fp32 rows whose magnitudes differ per shard; query validation under the query split (2 ranks) ---------------------
SC_N, SC_D = 2000, 48


def scale_data():
    rng = np.random.default_rng(31)
    xb = rng.standard_normal((SC_N, SC_D)).astype(np.float32)
    xb[:1000] *= 0.05   # shard 0
    xb[1000:] *= 40.0   # shard 1
    xb[1900] *= 3000.0  # far beyond the headroom of the exponent agreed from the shards' first rows (4 in this scenario)
    xq = rng.standard_normal((10, SC_D)).astype(np.float32)
    xq[7] *= 1.0e7      # leaves fp16's range under the index's scale: searched again with an exponent of its own
    return xb, xq


def _backend(backend_kind):
    if backend_kind == "hip":
        import torch
        from lotus_amd.backend import HipBackend

        torch.cuda.set_device(0)
        return HipBackend("cuda:0")
    from oracle_backend import OracleBackend

    return OracleBackend()


def worker_scale(rank, world, port, tmp, out_q, backend_kind):
    sys.path.insert(0, os.path.dirname(HERE))
    sys.path.insert(0, HERE)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch.distributed as dist

    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from lotus_amd import HipVS

        be = _backend(backend_kind)
        xb, xq = scale_data()
        res = {"rank": rank}
        HipVS._EXP_HEAD_ROWS = 4
        vs = HipVS(backend=be, shard=True)
        vs.index(None, xb, os.path.join(tmp, "sc"), persist=False)
        res["exp"] = vs._resident[vs.index_dir].packed.exp
        out = vs(xq[:6], 5)
        res["rows"] = (np.asarray(out.distances), np.asarray(out.indices))
        vq = HipVS(backend=be, shard="queries")
        vq.index(None, xb[:1000], os.path.join(tmp, "scq"), persist=False)
        out = vq(xq, 5)  # the out-of-range query sits in rank 1's slice; both ranks must take the retry
        res["queries"] = (np.asarray(out.distances), np.asarray(out.indices))
        bad = xq.copy()
        bad[2, 3] = np.nan  # rank 0's slice
        try:
            vq(bad, 5)
            res["nan"] = "no error"
        except ValueError as e:
            res["nan"] = str(e)
        bad = xb.copy()
        bad[1500, 0] = np.inf  # rank 1's shard
        try:
            HipVS(backend=be, shard=True).index(None, bad, os.path.join(tmp, "scbad"), persist=False)
            res["inf"] = "no error"
        except ValueError as e:
            res["inf"] = str(e)
        out_q.put(res)
        dist.barrier()
    finally:
        dist.destroy_process_group()


def run_scale(tmp, backend_kind):
    import torch.multiprocessing as mp

    world = 2
    port = 33500 + (os.getpid() % 2000) + (7 if backend_kind == "hip" else 0)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=worker_scale, args=(r, world, port, str(tmp), q, backend_kind)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=600) for _ in range(world)], key=lambda t: t["rank"])
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    return res


def check_scale(res, exact: bool):
    import synth
    from lotus_amd import HipVS
    from oracle_backend import OracleBackend

    xb, xq = scale_data()
    one = HipVS(backend=OracleBackend())  # the same rows on one rank: the exponent of the true maximum
    one.index(None, xb, "unused", persist=False)
    want_exp = one._resident[one.index_dir].packed.exp
    ref_rows = one(xq[:6], 5)
    one.index(None, xb[:1000], "unused2", persist=False)
    ref_q = one(xq, 5)

    def same(got, ref):
        D, I = got
        Dr, Ir = np.asarray(ref.distances), np.asarray(ref.indices)
        if exact:
            assert np.array_equal(I, Ir) and np.array_equal(D, Dr)
        else:
            scale = np.abs(Dr).max(axis=1, keepdims=True)
            err, hard, recall = synth.compare_topk(Dr / scale, Ir, D / scale, I, atol=1e-5)
            assert err <= 1e-5 and hard == 0 and recall >= 0.999, (err, hard, recall)

    for r in res:
        assert r["exp"] == want_exp, (r["exp"], want_exp)  # re-agreed after the outlier row: the exponent of the true maximum
        same(r["rows"], ref_rows)
        same(r["queries"], ref_q)
        assert "inf or NaN" in r["nan"] and "inf or NaN" in r["inf"], (r["nan"], r["inf"])
