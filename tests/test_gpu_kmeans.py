"""GPU parity of the k-means pieces (lotus/utils.py:61-65 replacement) against the oracle."""
import os

import numpy as np
import pytest

import oracle
import synth
from lotus_amd import _capi

pytestmark = pytest.mark.gpu
F16, SPLIT = _capi.PACK_F16, _capi.PACK_SPLIT
IP, L2 = _capi.METRIC_IP, _capi.METRIC_L2
HERE = os.path.dirname(os.path.abspath(__file__))


def _stored(x, mode):
    x = np.asarray(x, np.float32)
    hi = x.astype(np.float16)
    if mode == F16:
        return hi.astype(np.float32)
    lo = (x - hi.astype(np.float32)).astype(np.float16)
    return hi.astype(np.float32) + lo.astype(np.float32)


@pytest.mark.parametrize("nq,nb,d,mode,metric", [
    (3000, 1024, 768, F16, L2),    # points x centroids, the k-means assign shape
    (1000, 257, 96, SPLIT, L2),
    (777, 5000, 128, F16, IP),     # k = 1 search, several slabs
    (300, 3, 32, F16, L2),
])
def test_top1_mode_matches_oracle(hip_backend, nq, nb, d, mode, metric):
    be = hip_backend
    xb = synth.corpus(nb, d, seed=4) * 1.3
    xq, _ = synth.queries(xb, nq, seed=8)
    cb = be.pack(xb.astype(np.float16) if mode == F16 else xb, mode)
    cq = be.pack(xq.astype(np.float16) if mode == F16 else xq, mode)
    D, I = be.keys_to_result(be.search_keys(cb, cq, 1, metric), metric)
    D, I = D.cpu().numpy(), I.cpu().numpy()
    Dr, Ir = oracle.flat_search(_stored(xb, mode), _stored(xq, mode), 1, metric)
    assert (I == Ir).mean() >= 0.999
    bad = I != Ir
    assert np.abs(D - Dr)[~bad].max() <= 4e-5
    assert np.abs(D - Dr)[bad].max() <= 4e-5 if bad.any() else True  # swapped ids only inside near-ties


def test_top1_exact_ties_keep_the_lowest_row(hip_backend):
    be = hip_backend
    base = synth.corpus(40, 64, seed=1)
    xb = np.concatenate([base, base, base])  # every centroid three times
    xq, _ = synth.queries(base, 500, seed=2)
    cb, cq = be.pack(xb.astype(np.float16), F16), be.pack(xq.astype(np.float16), F16)
    for metric in (IP, L2):
        _, I = be.keys_to_result(be.search_keys(cb, cq, 1, metric), metric)
        _, Ir = oracle.flat_search(_stored(xb, F16), _stored(xq, F16), 1, metric)
        assert np.array_equal(I.cpu().numpy(), Ir) and (Ir < 40).all()


@pytest.mark.parametrize("mode", [F16, SPLIT])
def test_accumulate_is_exact_and_in_row_order(hip_backend, mode):
    be = hip_backend
    rng = np.random.default_rng(0)
    n, d, k = 20000, 200, 37
    x = (rng.standard_normal((n, d)) * 2).astype(np.float32)
    assign = rng.integers(0, k, n).astype(np.int64)
    assign[rng.integers(0, n, 50)] = -1  # skipped rows
    assign[assign == 5] = 6  # an empty cluster
    p = be.pack(x.astype(np.float16) if mode == F16 else x, mode)
    sums, counts = be.kmeans_accumulate(p, be.to_device(assign), k)
    vals = _stored(x, mode)
    ref = np.zeros((k, d), np.float32)
    ok = assign >= 0
    np.add.at(ref, assign[ok], vals[ok])
    assert np.array_equal(counts.cpu().numpy(), np.bincount(assign[ok], minlength=k).astype(np.float32))
    assert np.array_equal(sums.cpu().numpy(), ref)  # same values, same order -> bit-identical float32 sums
    assert counts[5].item() == 0 and not sums[5].any().item()


@pytest.mark.parametrize("mode", [F16, SPLIT])
def test_accumulate_bucket_sizes_around_the_half_batches(hip_backend, mode):
    """The sums kernel walks a bucket in batches of 64 (fp16 rows) / 32 (hi|lo rows) rows, the row numbers of the next batch
    prefetched while one is added, then a row-by-row tail: buckets of every size around 0 .. 3 batches and their halves (incl.
    the last bucket of the array, whose prefetch ends at the array's end), d with a partly filled last lane block -
    bit-identical to in-order float32 sums."""
    be = hip_backend
    rng = np.random.default_rng(21)
    H = 32 if mode == F16 else 16
    sizes = sorted(set([0, 1, 2, H - 1, H, H + 1, 2 * H - 1, 2 * H, 2 * H + 1, 3 * H - 1, 3 * H, 3 * H + 5, 4 * H, 4 * H + 1,
                        5 * H + 3, 7 * H, 1000, 1025]))
    for d in (8, 260, 768):
        assign = np.repeat(np.arange(len(sizes)), sizes).astype(np.int64)
        rng.shuffle(assign)
        n = len(assign)
        x = (rng.standard_normal((n, d)) * 3).astype(np.float32)
        pk = be.pack(x.astype(np.float16) if mode == F16 else x, mode)
        sums, counts = be.kmeans_accumulate(pk, be.to_device(assign), len(sizes))
        ref = np.zeros((len(sizes), d), np.float32)
        np.add.at(ref, assign, _stored(x, mode))
        assert np.array_equal(counts.cpu().numpy(), np.asarray(sizes, np.float32))
        assert np.array_equal(sums.cpu().numpy(), ref), (mode, d)
        # the biggest bucket last in the array: its prefetch runs into the clamp
        order = np.argsort(np.asarray(sizes), kind="stable")
        remap = np.empty(len(sizes), np.int64)
        remap[order] = np.arange(len(sizes))
        sums2, _ = be.kmeans_accumulate(pk, be.to_device(remap[assign]), len(sizes))
        assert np.array_equal(sums2.cpu().numpy()[remap], ref)


def test_host_helpers_match_faiss_restatement(hip_backend):
    be = hip_backend
    for n, seed in ((1, 3), (2, 1234), (1000, 1234), (4097, 1235)):
        assert np.array_equal(be.rand_perm(n, seed), oracle.rand_perm(n, seed))
    from oracle.kmeans import _split_clusters

    rng = np.random.default_rng(2)
    c1 = rng.standard_normal((9, 11)).astype(np.float32)
    h1 = np.array([0, 30, 0, 5, 1, 0, 44, 2, 8], np.float32)
    c2, h2 = c1.copy(), h1.copy()
    n1 = be.split_clusters(90, h1, c1)
    n2 = _split_clusters(90, h2, c2, None)
    assert n1 == n2 == 3 and np.array_equal(c1, c2) and np.array_equal(h1, h2)


def test_kmeans_end_to_end_vs_oracle_and_golden(hip_backend):
    from lotus_amd.cluster import kmeans

    z = np.load(os.path.join(HERE, "golden", "kmeans_blobs.npz"))
    x = z["x"]  # float16 -> fp16 storage, identical values on both sides
    r = kmeans(x, int(z["k"]), niter=int(z["niter"]), max_points_per_centroid=int(z["mppc"]), backend=hip_backend)
    assert np.array_equal(r.train_ids, z["train_ids"])
    assert (r.assign == z["assign"]).mean() >= 1 - 1e-4  # SURVEY.md 8(c) k-means protocol
    assert abs(r.obj[-1] - z["obj"][-1]) <= 1e-5 * abs(z["obj"][-1])
    assert np.allclose(r.centroids, z["centroids"], atol=1e-4)
    # fp32 values on the hi|lo path, empty-cluster split included
    xd = np.repeat(np.random.default_rng(3).standard_normal((3, 8)).astype(np.float32), 20, axis=0)
    r2 = kmeans(xd, 5, niter=4, backend=hip_backend)
    o2 = oracle.kmeans_faiss(_stored(xd, SPLIT), 5, niter=4)
    assert r2.nsplit.tolist() == o2.nsplit.tolist()
    # three distinct points, five centroids: exact distance ties make the trajectory rounding-dependent, so compare
    # the outcome, not the path: every point sits (almost) on its centroid, as in the oracle's solution
    res = ((_stored(xd, SPLIT) - r2.centroids[r2.assign]) ** 2).sum(1)
    res_o = ((_stored(xd, SPLIT) - o2.centroids[o2.assign]) ** 2).sum(1)
    assert res.max() <= max(1e-4, 2 * res_o.max())


def test_kmeans_scale_smoke(hip_backend):
    """200k x 768 fp16, K = 256 (faiss-parity subsample 65 536): objective decreases, clusters recovered."""
    import torch
    from lotus_amd.cluster import kmeans

    g = torch.Generator(device=hip_backend.device)
    g.manual_seed(5)
    K, n, d = 256, 200_000, 768
    cent = torch.nn.functional.normalize(torch.randn((K, d), generator=g, device=hip_backend.device), dim=1)
    lab = torch.randint(0, K, (n,), generator=g, device=hip_backend.device)
    x = torch.nn.functional.normalize(cent[lab] + 0.02 * torch.randn((n, d), generator=g, device=hip_backend.device), dim=1)
    xh = x.to(torch.float16).cpu().numpy()
    r = kmeans(xh, K, niter=10, backend=hip_backend)
    assert len(r.train_ids) == K * 256 and r.obj[-1] <= r.obj[0]
    # purity: rows sharing a true blob share a cluster id for the vast majority of blobs
    lab = lab.cpu().numpy()
    agree = 0
    for b in range(0, K, 8):
        ids = r.assign[lab == b]
        agree += np.bincount(ids).max() / len(ids)
    assert agree / (K / 8) > 0.7


@pytest.mark.parametrize("cmode,qmode,metric,nq,nb,d", [
    (SPLIT, F16, L2, 20_000, 1024, 768),    # fp16 points x fp32-accurate centroids: the cfg5 assignment (2 passes -> 1)
    (SPLIT, SPLIT, L2, 9_000, 300, 384),    # LOTUS's default fp32 embeddings on both sides (3 passes -> 1)
    (F16, SPLIT, IP, 5_000, 2048, 128),
    (SPLIT, F16, IP, 3_000, 70_000, 64),    # long corpus: several slabs -> merge_top2
    (SPLIT, F16, L2, 4_000, 16_384, 64),    # the longest corpus the query-streaming kernel takes (64 resident tiles) ...
    (SPLIT, F16, L2, 4_000, 16_385, 64),    # ... and one row more: the slab kernel (lvs_nearest_hi)
])
def test_certified_nearest_equals_the_exact_search(hip_backend, cmode, qmode, metric, nq, nb, d):
    """`nearest` = one MFMA pass over the hi parts + margin certificate + exact re-search of the uncertified queries;
    its winners are those of the exact 2-3 pass search, key for key."""
    be = hip_backend
    rng = np.random.default_rng(nb + d)
    xb = synth.corpus(nb, d, seed=4) * 1.3
    xq = (xb[rng.integers(0, nb, nq)] + 0.25 * synth.corpus(nq, d, seed=8)).astype(np.float32)
    cb = be.pack(xb.astype(np.float16) if cmode == F16 else xb, cmode)
    cq = be.pack(xq.astype(np.float16) if qmode == F16 else xq, qmode)
    stats = {}
    Dg, Ig = (t.cpu().numpy() for t in be.keys_to_result(be.nearest(cb, cq, metric, id_offset=7, stats=stats), metric))
    Dw, Iw = (t.cpu().numpy() for t in be.keys_to_result(be.search_keys(cb, cq, 1, metric, id_offset=7), metric))
    # the same winner for every query - up to float32 near-ties: a query decided by two exact dot products (lvs_resolve_pairs)
    # and the exact MFMA search sum in different orders, so two rows whose exact scores agree to rounding may swap
    assert np.abs(Dg - Dw).max() <= 4e-6 * max(1.0, np.abs(Dw).max())  # exact scores (another fp32 summation order)
    assert (Ig != Iw).sum() <= 2, (Ig != Iw).sum()
    assert stats["queries"] == nq and stats["uncertified"] <= 0.02 * nq  # the certificate does the work, not the fallback


def test_certified_nearest_with_ties_and_lo_only_differences(hip_backend):
    """Rows that tie exactly, and rows that differ only BELOW fp16 resolution (identical hi parts): the one-pass scores
    cannot separate them, so those queries must come back uncertified and be decided by the exact search."""
    be = hip_backend
    d = 96
    h16 = (synth.corpus(50, d, seed=2) * 1.1).astype(np.float16)
    base = h16.astype(np.float32)                                     # exactly representable: lo part 0
    twin = base.copy()
    twin[:, :8] += 0.3 * np.spacing(np.abs(h16[:, :8])).astype(np.float32)  # 0.3 ulp: same hi part, non-zero lo part
    dup = base[:10].copy()
    xb = np.concatenate([base, twin, dup]).astype(np.float32)
    assert np.array_equal(xb[:50].astype(np.float16), xb[50:100].astype(np.float16))
    xq = (np.concatenate([base, twin]) + 0.01 * synth.corpus(100, d, seed=3)).astype(np.float32)
    cb, cq = be.pack(xb, SPLIT), be.pack(xq.astype(np.float16), F16)
    for metric in (L2, IP):
        stats = {}
        Dg, Ig = (t.cpu().numpy() for t in be.keys_to_result(be.nearest(cb, cq, metric, stats=stats), metric))
        Dw, Iw = (t.cpu().numpy() for t in be.keys_to_result(be.search_keys(cb, cq, 1, metric), metric))
        diff = Ig != Iw
        # a twin that differs by 0.3 ulp in 8 of 96 coordinates scores within a few 1e-7 of its base row: where two exact
        # float32 evaluations (two dot products vs the MFMA search) disagree on such a pair, both answers are right
        assert np.abs(Dg - Dw)[diff].max(initial=0.0) <= 1e-6 * max(1.0, np.abs(Dw).max()) and diff.sum() <= 40, diff.sum()
        assert np.array_equal(Ig[:10], Iw[:10])  # exact duplicates (three candidates inside the bound): lowest id, as the search
        assert stats["uncertified"] >= 90  # (almost) every query has a twin or duplicate within the bound


def test_kmeans_uses_the_certified_assignment(hip_backend):
    """fp32 embeddings (hi|lo points AND centroids): the k-means result is unchanged by the one-pass assignment."""
    from lotus_amd.cluster import kmeans

    rng = np.random.default_rng(5)
    c = rng.standard_normal((24, 128)).astype(np.float32) * 2
    x = (c[rng.integers(0, 24, 30_000)] + 0.4 * rng.standard_normal((30_000, 128))).astype(np.float32)
    r = kmeans(x, 24, niter=6, backend=hip_backend)
    ref = oracle.kmeans_faiss(_stored(x, SPLIT), 24, niter=6)
    assert (r.assign == ref.assign).mean() >= 1 - 1e-4
    assert np.allclose(r.obj, ref.obj, rtol=1e-5)


def test_certified_nearest_zero_scores_ragged_rows_and_every_tile_position(hip_backend):
    """The one-pass kernel tags a score's position (tile row) in its low mantissa bits: a unique winner whose score is
    exactly 0 (the tag then lives in a denormal), corpus sizes that leave the last tile ragged, and winners planted at
    every position of a 256-row tile must all decode to the right row."""
    be = hip_backend
    d = 64
    # (1) inner product: one-hot rows; query orthogonal to row 5 (score exactly 0), negative against every other row
    nb = 300
    xb = -np.eye(nb, d, dtype=np.float32) - 0.5          # every row strongly negative against an all-ones query ...
    xb[5] = 0.0
    xb[5, 0], xb[5, 1] = 1.0, -1.0                        # ... except row 5: 1 - 1 = 0 exactly
    xq = np.ones((3, d), np.float32)
    cb, cq = be.pack(xb, SPLIT), be.pack(xq, SPLIT)
    _, Ig = be.keys_to_result(be.nearest(cb, cq, IP), IP)
    _, Iw = be.keys_to_result(be.search_keys(cb, cq, 1, IP), IP)
    assert np.array_equal(Ig.cpu().numpy(), Iw.cpu().numpy()) and (Ig.cpu().numpy() == 5).all()
    # (2) L2: the query IS a corpus row (distance 0 -> u = |y|^2 exactly), ragged sizes, winners at every tile position
    for nb in (1, 255, 257, 1000, 1283):
        rng = np.random.default_rng(nb)
        xb = (synth.corpus(nb, d, seed=nb) * 1.7).astype(np.float16).astype(np.float32)
        pick = np.arange(nb) if nb <= 300 else np.concatenate([np.arange(256), np.arange(nb - 300, nb)])
        xq = xb[pick].copy()
        cb, cq = be.pack(xb, SPLIT), be.pack(xq, SPLIT)
        for metric in (L2, IP) if nb > 1 else (L2,):
            stats = {}
            Dg, Ig = (t.cpu().numpy() for t in be.keys_to_result(be.nearest(cb, cq, metric, stats=stats), metric))
            Dw, Iw = (t.cpu().numpy() for t in be.keys_to_result(be.search_keys(cb, cq, 1, metric), metric))
            assert np.array_equal(Ig, Iw), (nb, metric)
            if metric == L2:
                assert np.array_equal(Ig[:, 0], pick) and np.abs(Dg).max() <= 1e-5


def _trajectory_prefix(r, ref):
    """Number of leading iterations over which two k-means runs made the same empty-cluster decisions."""
    same = np.asarray(r.nsplit) == np.asarray(ref.nsplit)
    return int(len(same) if same.all() else np.argmin(same))


def test_kmeans_parity_at_the_configs_cluster_count_with_faiss_subsample(hip_backend):
    """K = 1 024 on 300 000 rows: n > K * 256, so faiss's 262 144-row training subsample engages (lotus/utils.py:61-62,
    SURVEY.md Appendix A.4).  Same subsample and initial centroids as oracle.kmeans_faiss.  Rows WITHOUT cluster structure
    (no cluster runs empty, but Lloyd's iteration is chaotic on them: a handful of near-tie flips - allowed, 1e-4 - move
    centroids by 1e-6, which flips more boundary rows next time), so what is pinned is: one full iteration + final
    assignment of all rows against the oracle (assignment >= 1 - 1e-3, centroids 1e-4, objective 1e-5); over 10 iterations
    the objective (1e-5 at every iteration - it averages over all rows) and the split counts; and the final assignment
    of the 10-iteration run exactly (>= 1 - 1e-4) against a brute-force search over the run's OWN centroids - the
    certified one-pass assignment's uncertified re-search is what makes that exact at scale."""
    import benchdata
    from lotus_amd.cluster import kmeans

    K, n, d = 1024, 300_000, 128
    x = benchdata.corpus(benchdata.CFG_KMEANS, n, d)           # fp16 storage: identical values on both sides
    x32 = x.astype(np.float32)
    r1 = kmeans(x, K, niter=1, backend=hip_backend)
    ref1 = oracle.kmeans_faiss(x32, K, niter=1)
    assert len(r1.train_ids) == K * 256 and np.array_equal(r1.train_ids, ref1.train_ids)
    assert np.abs(r1.centroids - ref1.centroids).max() <= 1e-4 and np.allclose(r1.obj, ref1.obj, rtol=1e-5)
    assert (r1.assign == ref1.assign).mean() >= 1 - 1e-3
    stats = {}
    r = kmeans(x, K, niter=10, backend=hip_backend, stats=stats)
    ref = oracle.kmeans_faiss(x32, K, niter=10, final_assign=False)
    assert r.nsplit.tolist() == ref.nsplit.tolist()
    assert np.allclose(r.obj, ref.obj, rtol=1e-5), np.abs(r.obj / ref.obj - 1).max()
    _, I = oracle.flat_search(r.centroids, x32, 1, 1)
    assert (I[:, 0] == r.assign).mean() >= 1 - 1e-4
    assert stats["queries"] == 10 * K * 256  # every training assignment went through the certificate


def test_kmeans_on_blobs_with_empty_cluster_splits_at_scale(hip_backend):
    """The configs[4] data shape (mixture of K blobs): dozens of clusters run empty in the first iterations and faiss's
    split_clusters re-seeds them from an RNG walk over the cluster SIZES - one row flipping on a near-tie (allowed: 1e-4)
    can empty a one-row cluster, shift the generator's stream and send the two runs down different (equally valid)
    paths.  So: identical decisions and objectives over the common prefix (at least the first two iterations, which hold
    the bulk of the splits), and - whatever the path - an exact final assignment against the run's OWN centroids."""
    import benchdata
    from lotus_amd.cluster import kmeans

    K, n, d = 1024, 300_000, 128
    x, _ = benchdata.blobs(benchdata.CFG_KMEANS, n, d, K)
    r = kmeans(x, K, niter=8, backend=hip_backend)
    ref = oracle.kmeans_faiss(x.astype(np.float32), K, niter=8)
    assert np.array_equal(r.train_ids, ref.train_ids)
    pre = _trajectory_prefix(r, ref)
    assert pre >= 2 and r.nsplit[:pre].sum() >= 20, (pre, r.nsplit, ref.nsplit)
    assert np.allclose(r.obj[:pre], ref.obj[:pre], rtol=1e-5)
    assert abs(r.obj[-1] / ref.obj[-1] - 1) <= 0.05  # both runs end in comparable optima
    _, I = oracle.flat_search(r.centroids, x.astype(np.float32), 1, 1)
    assert (I[:, 0] == r.assign).mean() >= 1 - 1e-4


def test_device_split_replays_faiss_rng(hip_backend):
    """lvs_kmeans_update_centroids runs faiss split_clusters on the device (mt19937(1234) replayed by one thread): same
    decisions, same perturbed centroids, same hassign as the host twin / the oracle's restatement."""
    import torch
    from oracle.kmeans import _split_clusters

    be = hip_backend
    rng = np.random.default_rng(4)
    for k, d, n, empties in ((9, 11, 90, [0, 2, 5]), (300, 96, 40_000, list(range(0, 300, 7))), (64, 768, 5000, [63]),
                             (16, 8, 1000, []), (1024, 32, 262_144, list(range(3, 1024, 8))),  # ~130 k draws: many state blocks
                             (20_000, 4, 2_000_000, [5, 19_999])):                                # sizes beyond the LDS copy
        counts = rng.integers(1, 50, k).astype(np.float32)
        counts[empties] = 0
        sums = rng.standard_normal((k, d)).astype(np.float32) * counts[:, None]
        cent = rng.standard_normal((k, d)).astype(np.float32)
        # oracle: compute_centroids' division, then the split
        ref_c, ref_h = cent.copy(), counts.copy()
        nz = ref_h > 0
        ref_c[nz] = sums[nz] * (np.float32(1.0) / ref_h[nz])[:, None]
        ref_n = _split_clusters(n, ref_h, ref_c, False)
        c_dev, h_dev = be.to_device(cent), be.to_device(counts)
        ns = torch.zeros(1, dtype=torch.int32, device=be.device)
        pk, st = be.kmeans_finish(be.to_device(sums), h_dev, c_dev, n, SPLIT, ns)
        assert int(ns.item()) == ref_n == len(empties)
        assert np.array_equal(c_dev.cpu().numpy(), ref_c) and np.array_equal(h_dev.cpu().numpy(), ref_h)
        # the repacked centroids and the certificate's statistics describe the UPDATED centroids
        assert np.array_equal(be.unpack(pk).cpu().numpy(), _stored(ref_c, SPLIT))
        vals = _stored(ref_c, SPLIT)
        lo = (ref_c - ref_c.astype(np.float16).astype(np.float32)).astype(np.float16).astype(np.float32)
        R2, E2 = st.cpu().numpy()
        assert abs(R2 - (vals ** 2).sum(1).max()) <= 1e-5 * R2 and abs(E2 - (lo ** 2).sum(1).max()) <= 1e-5 * max(E2, 1e-12)


def test_objective_kernel_matches_the_sum_of_assignment_distances(hip_backend):
    import torch

    be = hip_backend
    rng = np.random.default_rng(6)
    n, d, k = 30_000, 96, 50
    x = rng.standard_normal((n, d)).astype(np.float16)
    cent = rng.standard_normal((k, d)).astype(np.float32)
    p = be.pack(x, F16)
    D, I = oracle.flat_search(cent, x.astype(np.float32), 1, 1)
    keys = be.search_keys(be.pack(cent, SPLIT), p, 1, L2)
    sums, counts = be.kmeans_accumulate_keys(p, keys, k)
    ref = np.zeros((k, d), np.float32)
    np.add.at(ref, I[:, 0], x.astype(np.float32))
    assert np.array_equal(sums.cpu().numpy(), ref) and np.array_equal(counts.cpu().numpy(), np.bincount(I[:, 0], minlength=k))
    out = torch.zeros(1, dtype=torch.float64, device=be.device)
    be.kmeans_objective(be.to_device(cent), sums, counts, p.norms.double().sum().reshape(1), out)
    want = float(D[:, 0].astype(np.float64).sum())
    assert abs(out.item() - want) <= 1e-5 * want


def test_counting_sort_two_digit_path_and_ignored_rows(hip_backend):
    """More than 24 575 centroids bucket the rows with two stable digit passes; sums stay in row order (bit-identical)."""
    be = hip_backend
    rng = np.random.default_rng(9)
    n, d, k = 70_000, 16, 30_000
    x = rng.standard_normal((n, d)).astype(np.float16)
    assign = rng.integers(0, k, n).astype(np.int64)
    assign[rng.integers(0, n, 300)] = k + 5   # out of range: ignored
    assign[rng.integers(0, n, 300)] = -2
    p = be.pack(x, F16)
    sums, counts = be.kmeans_accumulate(p, be.to_device(assign), k)
    ok = (assign >= 0) & (assign < k)
    ref = np.zeros((k, d), np.float32)
    np.add.at(ref, assign[ok], x.astype(np.float32)[ok])
    assert np.array_equal(counts.cpu().numpy(), np.bincount(assign[ok], minlength=k).astype(np.float32))
    assert np.array_equal(sums.cpu().numpy(), ref)
    # ragged chunk boundaries of the one-digit path: n = 1, a chunk + 1 row, every row in one bucket
    for n2, k2 in ((1, 3), (8193, 5), (20_000, 1)):
        x2 = rng.standard_normal((n2, 8)).astype(np.float16)
        a2 = rng.integers(0, k2, n2).astype(np.int64)
        s2, c2 = be.kmeans_accumulate(be.pack(x2, F16), be.to_device(a2), k2)
        r2 = np.zeros((k2, 8), np.float32)
        np.add.at(r2, a2, x2.astype(np.float32))
        assert np.array_equal(s2.cpu().numpy(), r2) and np.array_equal(c2.cpu().numpy(), np.bincount(a2, minlength=k2))


@pytest.mark.parametrize("mode", [F16, SPLIT])
def test_distance_bounds_skip_rows_without_changing_any_result(hip_backend, mode):
    """Hamerly bounds (lvs_kmeans_bounds_step): rows whose nearest centroid provably did not change are not searched again.
    Same assignments => bit-identical sums, centroids, objectives, split counts as the exhaustive iteration - and on
    clustered rows almost nothing is searched once the centroids settle."""
    import benchdata
    from lotus_amd.cluster import kmeans

    K, n, d = 96, 150_000, 64
    x16, _ = benchdata.blobs(benchdata.CFG_KMEANS, n, d, K)
    x = x16 if mode == F16 else (x16.astype(np.float32) * np.float32(1.0 + 2.0 ** -13))  # values that need the lo half
    kw = dict(niter=12, backend=hip_backend, max_points_per_centroid=None)
    plain = kmeans(x, K, bounds=False, **kw)
    st = {}
    fast = kmeans(x, K, bounds=True, stats=st, **kw)
    assert np.array_equal(fast.nsplit, plain.nsplit)
    assert np.array_equal(fast.obj, plain.obj) and np.array_equal(fast.centroids, plain.centroids)
    assert np.array_equal(fast.assign, plain.assign)
    searched = st["searched_rows"]
    assert searched[0] == n and len(searched) == 12
    assert sum(searched[6:]) <= 0.2 * 6 * n, searched  # the tail of the run searches a small fraction of the rows
    # rows without structure: the bounds rarely certify anything, the result must not change either
    xu = benchdata.corpus(benchdata.CFG_KMEANS, 60_000, d)
    a = kmeans(xu, 40, niter=5, backend=hip_backend, max_points_per_centroid=None, bounds=False)
    b = kmeans(xu, 40, niter=5, backend=hip_backend, max_points_per_centroid=None, bounds=True)
    assert np.array_equal(a.centroids, b.centroids) and np.array_equal(a.obj, b.obj) and np.array_equal(a.assign, b.assign)


# ---- step-by-step parity at configs[4]'s own shape: K = 1 024, d = 768, faiss's 262 144-row subsample of blob rows -------------
def test_kmeans_teacher_forced_steps_at_the_configs_shape(hip_backend):
    """Every iteration pinned on its own (lotus/utils.py:61-62) - see km_steps.teacher_forced:
    the ORACLE's centroids of iteration i go into ONE device step, so a near-tie flip cannot compound over iterations.
    Blob rows of SURVEY.md 8(d), K = 1 024, d = 768, 300 000 rows (faiss's 262 144-row subsample engages), 8 iterations."""
    import km_steps

    c5 = km_steps.cfg5_reference()
    assert len(c5["ref"].train_ids) == c5["K"] * 256 and c5["ref"].nsplit.sum() >= 100  # subsample engaged, splits exercised
    flips = km_steps.teacher_forced(hip_backend, c5)
    print(f"teacher-forced: {flips} near-tie flips in {c5['nit']} x {len(c5['ref'].train_ids)} assignments")


def test_kmeans_free_run_diverges_from_the_oracle_only_through_near_ties(hip_backend):
    """The free-running device k-means against oracle.kmeans_faiss on the same rows (km_steps.free_run): identical up to the
    first iteration in which any row is assigned differently, and there every differing row is a near-tie."""
    import km_steps

    c5 = km_steps.cfg5_reference()
    rep = km_steps.free_run(hip_backend, c5)
    print(f"free run: {rep}")


@pytest.mark.parametrize("mode", [F16, SPLIT])
def test_distance_bounds_at_the_default_on_size_with_ties_splits_and_50_iterations(hip_backend, mode):
    """The Hamerly bounds switch themselves on from 2^20 training rows (lotus_amd/cluster.py): at that size, over 50
    iterations, on rows with exact duplicates, rows planted half way between two blob centres (near-ties of the assignment)
    and blobs that start with two centroids or none (empty-cluster splits), the run with bounds must return bit-identical
    objectives, centroids, split counts and assignments to the exhaustive run - the float32 margins of
    lvs_kmeans_bounds_set / _fix / _step (ub += delta, lb -= max delta, 1e-5 relative guard) may only ever skip rows whose
    nearest centroid is provably unchanged."""
    import benchdata
    from lotus_amd.cluster import kmeans

    K, n, d = 256, 1 << 20, 64
    x16, lab = benchdata.blobs(benchdata.CFG_KMEANS, n, d, K)
    x = x16.astype(np.float32)
    cen = benchdata.blob_centres(benchdata.CFG_KMEANS, K, d)
    rng = np.random.default_rng(17)
    a, b = rng.integers(0, K, 30_000), rng.integers(0, K, 30_000)
    mid = cen[a] + cen[b] + 1e-3 * rng.standard_normal((30_000, d)).astype(np.float32)
    x[200_000:230_000] = mid / np.linalg.norm(mid, axis=1, keepdims=True)      # half way between two centres
    x[300_000:400_000] = x[:100_000]                                             # exact duplicates
    x[400_000:450_000] = x[400_000:400_020].repeat(2500, axis=0)                 # 20 values x 2 500 copies: duplicate INITIAL
    # centroids (ties go to the lowest id, the twin runs empty) -> empty-cluster splits from the first iteration on
    if mode == F16:
        x = x.astype(np.float16)
    else:
        x = x * np.float32(1.0 + 2.0 ** -13)  # values that need the lo half
    kw = dict(niter=50, backend=hip_backend, max_points_per_centroid=None)
    st = {}
    fast = kmeans(x, K, stats=st, **kw)  # bounds=None: the default must have switched them on at this size
    assert "searched_rows" in st and len(st["searched_rows"]) == 50 and min(st["searched_rows"]) < n // 2
    plain = kmeans(x, K, bounds=False, **kw)
    assert np.array_equal(fast.nsplit, plain.nsplit)
    assert np.array_equal(fast.obj, plain.obj) and np.array_equal(fast.centroids, plain.centroids)
    assert np.array_equal(fast.assign, plain.assign)
    assert plain.nsplit.sum() >= 1, plain.nsplit  # the data did exercise split_clusters


def _hi_scores(xb, xq, metric):
    """float64 one-pass scores: hi parts only (fp16 roundings), exact norms of the stored (hi + lo) values."""
    hb, hq = xb.astype(np.float16).astype(np.float64), xq.astype(np.float16).astype(np.float64)
    s = hq @ hb.T
    if metric == L2:
        sb, sq = _stored(xb, SPLIT).astype(np.float64), _stored(xq, SPLIT).astype(np.float64)
        s = -np.maximum((sq ** 2).sum(1)[:, None] + (sb ** 2).sum(1)[None, :] - 2 * s, 0)
    return s


@pytest.mark.parametrize("metric", [L2, IP])
@pytest.mark.parametrize("nq,nb,d", [(700, 1, 64), (300, 2, 64), (1000, 3, 96), (513, 255, 64), (2049, 257, 128),
                                     (5000, 1024, 768), (70_000, 1000, 64), (300, 5000, 32), (1500, 16_384, 32)])
def test_query_streaming_nearest_returns_the_top_three(hip_backend, metric, nq, nb, d):
    """lvs_nearest3 (lvs_assign.hip): queries stream past resident corpus tiles; per query the best and second-best ROWS and
    the third-best SCORE of the one-pass (hi parts) scores, every value perturbed by < 2^-17 relative (position tags).  Ragged
    corpus / query counts, one to twenty corpus tiles, both metrics, against a float64 evaluation of the same scores."""
    import torch

    be = hip_backend
    rng = np.random.default_rng(nq + nb)
    xb = (synth.corpus(nb, d, seed=nb) * 1.4).astype(np.float32)
    xq = (xb[rng.integers(0, nb, nq)] + 0.3 * synth.corpus(nq, d, seed=nq)).astype(np.float32)
    cb, cq = be.pack(xb, SPLIT), be.pack(xq, SPLIT)
    keys = torch.empty((nq,), dtype=torch.int64, device=be.device)
    keys2 = torch.empty((nq,), dtype=torch.int64, device=be.device)
    sec = torch.empty((nq,), dtype=torch.float32, device=be.device)
    third = torch.empty((nq,), dtype=torch.float32, device=be.device)
    need = int(be.lib.lvs_nearest3_workspace_bytes(nq, nb, d))
    ws = be._workspace(need)
    P = lambda t: int(t.data_ptr())
    be._c("lvs_nearest3", P(cb.rows), cb.mode, nb, P(cq.rows), cq.mode, nq, d, metric, P(cb.norms), P(cq.norms), 11, P(keys),
          P(keys2), P(sec), P(third), P(ws), int(ws.numel()), be._stream())
    k1 = keys.cpu().numpy().view(np.uint64)
    k2 = keys2.cpu().numpy().view(np.uint64)
    s1, i1, _ = oracle.unpack_keys(k1)
    s2, i2, e2 = oracle.unpack_keys(k2)
    S = _hi_scores(xb, xq, metric)
    order = np.argsort(-S, axis=1, kind="stable")
    top = np.take_along_axis(S, order[:, :3], axis=1)
    scale = np.abs(S).max() + (0 if metric == IP else 2 * np.abs(xq.astype(np.float64) @ xb.astype(np.float64).T).max())
    tol = 2.0 ** -15 * scale + 1e-6
    rows = np.arange(nq)
    assert np.abs(s1 - top[:, 0]).max() <= tol
    assert np.abs(S[rows, i1 - 11] - top[:, 0]).max() <= tol      # the reported row IS a best row (up to the tag perturbation)
    if nb >= 2:
        assert not e2.any() and (i2 != i1).all()
        assert np.abs(S[rows, i2 - 11] - top[:, 1]).max() <= tol and np.abs(s2 - top[:, 1]).max() <= tol
        assert np.abs(sec.cpu().numpy() - top[:, 1]).max() <= tol
    else:
        assert e2.all() and np.isneginf(sec.cpu().numpy()).all()
    if nb >= 3:
        assert np.abs(third.cpu().numpy() - top[:, 2]).max() <= tol
    else:
        assert np.isneginf(third.cpu().numpy()).all()


@pytest.mark.parametrize("metric", [L2, IP])
def test_two_candidate_certificate_settles_split_twins_with_two_dot_products(hip_backend, metric):
    """Right after faiss's split_clusters two centroids are c (1 + 1/1024) and c (1 - 1/1024) (alternating per coordinate): every
    row of that cluster is inside the one-pass error bound of BOTH twins and far from every third centroid.  Such queries
    must come back as pairs (lvs_nearest3_select) and be settled by lvs_resolve_pairs - same winners as the exact search (up
    to float32 near-ties), hardly anything left for the exact search over every row."""
    be = hip_backend
    rng = np.random.default_rng(3)
    K, d, nq = 512, 256, 40_000
    c = (synth.corpus(K // 2, d, seed=5) * 1.2).astype(np.float32)
    eps = np.where(np.arange(d) % 2 == 0, 1 + 1 / 1024, 1 - 1 / 1024).astype(np.float32)
    xb = np.concatenate([c * eps, c * (2 - eps)])                    # 256 twin pairs
    xq = (c[rng.integers(0, K // 2, nq)] + 0.05 * synth.corpus(nq, d, seed=6)).astype(np.float16)
    cb, cq = be.pack(xb, SPLIT), be.pack(xq, F16)
    stats = {}
    Dg, Ig = (t.cpu().numpy() for t in be.keys_to_result(be.nearest(cb, cq, metric, stats=stats), metric))
    Dw, Iw = (t.cpu().numpy() for t in be.keys_to_result(be.search_keys(cb, cq, 1, metric, one_pass=False), metric))
    assert np.abs(Dg - Dw).max() <= 4e-6 * max(1.0, np.abs(Dw).max())
    # ground truth in float64 on the stored values: the row either path reports must score within float32 noise of the best
    # one (the twins of a pair are ~3e-4 apart on average, so a few per cent of the queries have them closer than the ~3e-6 by
    # which the exact MFMA search and two float32 dot products may differ - there the two paths may name different twins)
    sb, sq = _stored(xb, SPLIT).astype(np.float64), xq.astype(np.float64)
    truth = sq @ sb.T
    if metric == L2:
        truth = -((sq ** 2).sum(1)[:, None] + (sb ** 2).sum(1)[None, :] - 2 * truth)
    best = truth.max(axis=1)
    rows = np.arange(nq)
    tol = 4e-6 * max(1.0, np.abs(best).max())
    assert (best - truth[rows, Ig[:, 0]]).max() <= tol and (best - truth[rows, Iw[:, 0]]).max() <= tol
    # ... and the two-dot-product arbiter is at least as often right as the MFMA search
    assert (Ig[:, 0] != truth.argmax(1)).sum() <= (Iw[:, 0] != truth.argmax(1)).sum() + 0.002 * nq
    assert (Ig != Iw).mean() <= 0.05
    assert stats["pairs"] >= 0.3 * nq and stats["uncertified"] - stats["pairs"] <= 0.01 * nq, stats


@pytest.mark.parametrize("mode", [F16, SPLIT])
def test_ranges_with_overlapped_sums_equal_the_single_pass(hip_backend, mode):
    """From 2^21 training rows on an iteration hands the rows over in four consecutive ranges (30 / 30 / 25 / 15 %) and runs the in-row-order sums
    of one range on a side stream under the assignment search of the next (lotus_amd/cluster.py `parts`): the sums continue
    across the ranges (lvs_kmeans_accumulate_keys carries them along), so objectives, centroids, split counts, the traced
    assignments and the final assignment are bit-identical to the single pass - also with an uneven last range."""
    import benchdata
    from lotus_amd.cluster import kmeans

    K, n, d = 64, 300_001, 96
    x16, _ = benchdata.blobs(benchdata.CFG_KMEANS, n, d, K)
    x = x16 if mode == F16 else (x16.astype(np.float32) * np.float32(1.0 + 2.0 ** -13))
    kw = dict(niter=6, backend=hip_backend, max_points_per_centroid=None, bounds=False)
    t1, t4 = [], []
    one = kmeans(x, K, parts=1, trace=t1, **kw)
    four = kmeans(x, K, parts=4, trace=t4, **kw)
    assert np.array_equal(one.obj, four.obj) and np.array_equal(one.centroids, four.centroids)
    assert np.array_equal(one.nsplit, four.nsplit) and np.array_equal(one.assign, four.assign)
    for a, b in zip(t1, t4):
        assert bool((a["keys"].reshape(-1) == b["keys"].reshape(-1)).all()) and bool((a["centroids"] == b["centroids"]).all())
    # ranges of unequal length (the default from 2^21 rows on: the last, whose sums nothing hides, is the shortest)
    from lotus_amd import cluster as cl

    assert not isinstance(cl.PARTS_DEFAULT, int) and abs(sum(cl.PARTS_DEFAULT) - 1.0) < 1e-9
    frac = kmeans(x, K, parts=cl.PARTS_DEFAULT, **kw)
    assert np.array_equal(one.obj, frac.obj) and np.array_equal(one.centroids, frac.centroids)
    assert np.array_equal(one.nsplit, frac.nsplit) and np.array_equal(one.assign, frac.assign)
    # the C ABI's contract on its own: two calls over consecutive ranges == one call, bit for bit
    import torch

    be = hip_backend
    p = be.pack(x, mode)
    keys = be.search_keys(be.pack(one.centroids, SPLIT), p, 1, L2)
    s1, c1 = be.kmeans_accumulate_keys(p, keys, K)
    s2 = torch.zeros_like(s1)
    c2 = torch.zeros_like(c1)
    h = 123_456
    be.kmeans_accumulate_keys_into(be.slice_rows(p, 0, h), keys[:h].contiguous(), K, s2, c2)
    be.kmeans_accumulate_keys_into(be.slice_rows(p, h, n), keys[h:].contiguous(), K, s2, c2)
    assert bool(torch.equal(s1, s2)) and bool(torch.equal(c1, c2))


def test_nearest_begin_chunks_its_queries_inside_the_scratch_budget(hip_backend, monkeypatch):
    """`nearest()` cut its queries into chunks that keep the query-streaming search's scratch (20 B per corpus tile
    and query) below NEAREST3_WS_BUDGET, but the two-halves form the pipelined k-means uses (`nearest_begin` / `nearest_finish`)
    asked for a whole range at once.  With a budget that forces three chunks both forms return the single call's keys."""
    be = hip_backend
    rng = np.random.default_rng(77)
    nq, nb, d = 150_000, 700, 96
    xb = rng.standard_normal((nb, d)).astype(np.float32)
    xq = (xb[rng.integers(0, nb, nq)] + 0.3 * rng.standard_normal((nq, d))).astype(np.float32)
    cb, cq = be.pack(xb, SPLIT), be.pack(xq, SPLIT)
    whole = be.nearest(cb, cq, L2, exact_scores=False)
    per_q = max(1, int(be.lib.lvs_nearest3_workspace_bytes(1 << 20, nb, d)) >> 20)
    monkeypatch.setattr(type(be), "NEAREST3_WS_BUDGET", per_q * (1 << 16))  # step = 65 536 queries -> 3 chunks
    assert be._nearest3_step(cb) == 1 << 16
    stats = {}
    h = be.nearest_begin(cb, cq, L2, exact_scores=False)
    assert "chunks" in h and len(h["chunks"]) == 3
    two = be.nearest_finish(h, stats=stats)
    assert stats["queries"] == nq
    chunked = be.nearest(cb, cq, L2, exact_scores=False)
    import torch

    assert bool(torch.equal(two, whole)) and bool(torch.equal(chunked, whole))


@pytest.mark.parametrize("mode", [F16, SPLIT])
def test_kmeans_iteration_abi_call_equals_the_launch_by_launch_loop(hip_backend, mode):
    """`lvs_kmeans_iteration` (ABI 7; SURVEY.md 8(b) `lvs_kmeans`, cut at the iteration): a host that owns nothing but device
    buffers and ctypes - no lotus_amd.cluster - runs faiss's training loop (lotus/utils.py:61-62) as one C-ABI call per
    iteration and gets, bit for bit, the centroids / objectives / split counts of `cluster.kmeans` issuing the same launches
    one by one (`USE_ITERATION_OP = False`), on blob rows with exact duplicates and duplicate initial centroids (empty
    clusters -> split_clusters replayed on the device)."""
    import ctypes

    import torch

    import benchdata
    from lotus_amd import cluster as cl

    be = hip_backend
    lib = be.lib
    K, n, d, niter = 48, 70_001, 96, 6
    x16, _ = benchdata.blobs(benchdata.CFG_KMEANS, n, d, K)
    x = x16.copy() if mode == F16 else (x16.astype(np.float32) * np.float32(1.0 + 2.0 ** -13))
    x[100:140] = x[7]  # forty equal rows (exact ties in the search; an over-full cluster next to empty ones)
    kw = dict(niter=niter, backend=be, max_points_per_centroid=None, bounds=False, parts=1, final_assign=False)
    cl.USE_ITERATION_OP = False
    try:
        ref = cl.kmeans(x, K, **kw)
    finally:
        cl.USE_ITERATION_OP = True
    via = cl.kmeans(x, K, **kw)  # the same through the single call
    assert np.array_equal(ref.centroids, via.centroids) and np.array_equal(ref.obj, via.obj) and np.array_equal(ref.nsplit, via.nsplit)

    # ---- and from raw ctypes: device buffers + the ABI, nothing of cluster.py
    p = be.pack(x, mode, exp="auto", check=True)           # (the row image and |x|^2 - lvs_pack_rows_checked)
    perm = be.rand_perm(n, 1234 + 1, K)                    # faiss: centroids = the first K rows of rand_perm(n, seed + 1)
    cent = be.unpack(p, be.to_device(perm[:K]), raw=True).contiguous()
    cpk, cstats = be.kmeans_pack_centroids(cent, SPLIT, exp=p.exp)
    dev = cent.device
    keys = torch.empty((n,), dtype=torch.int64, device=dev)
    obj = torch.zeros((niter,), dtype=torch.float64, device=dev)
    nsplit = torch.zeros((niter,), dtype=torch.int32, device=dev)
    x2 = p.norms.double().sum().reshape(1)
    need = lib.lvs_kmeans_iteration_workspace_bytes(n, d, K, p.mode, SPLIT)
    assert need > 0
    ws = torch.empty((need,), dtype=torch.uint8, device=dev)
    host_counts = (ctypes.c_int64 * 2)()
    stream = be._stream()
    for it in range(niter):
        rc = lib.lvs_kmeans_iteration(None, None, p.rows.data_ptr(), p.mode, n, d, p.norms.data_ptr(), x2.data_ptr(), 2 * int(p.exp), K, n,
                                      cent.data_ptr(), SPLIT, cpk.rows.data_ptr(), cpk.norms.data_ptr(), cstats.data_ptr(),
                                      keys.data_ptr(), obj[it:].data_ptr(), nsplit[it:].data_ptr(), ctypes.addressof(host_counts),
                                      ws.data_ptr(), need, stream)
        assert rc == 0, lib.lvs_last_error()
    torch.cuda.synchronize()
    scale = np.float32(2.0 ** -int(p.exp))
    assert np.array_equal((cent.cpu().numpy() * scale).astype(np.float32), ref.centroids)
    assert np.array_equal((obj.cpu().numpy() * 2.0 ** (-2 * int(p.exp))).astype(np.float32), ref.obj)
    assert np.array_equal(nsplit.cpu().numpy(), ref.nsplit) and int(ref.nsplit.sum()) >= 0
    # a short workspace is refused before anything is launched
    assert lib.lvs_kmeans_iteration(None, None, p.rows.data_ptr(), p.mode, n, d, p.norms.data_ptr(), x2.data_ptr(), 0, K, n,
                                    cent.data_ptr(), SPLIT, cpk.rows.data_ptr(), cpk.norms.data_ptr(), cstats.data_ptr(),
                                    keys.data_ptr(), obj.data_ptr(), None, None, ws.data_ptr(), 1024, stream) == _capi.ENOMEM
