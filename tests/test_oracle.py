"""Pins the oracle: hand-checkable cases, agreement of its independent implementations (numpy/BLAS blocks,
plain-C twin, float64 brute force), and the published behaviours of faiss it restates."""
import numpy as np
import pytest

import oracle
import synth
from oracle import cbind
from oracle.flat import FLT_MAX, _finish, _ord32, _unord32


def test_orthonormal_basis_is_hand_checkable():
    xb = np.eye(6, dtype=np.float32)
    xq = np.array([[0, 0, 3, 0, 0, 0], [1, 2, 0, 0, 0, 0]], np.float32)
    D, I = oracle.flat_search(xb, xq, 3)
    assert I[0].tolist() == [2, 0, 1] and D[0].tolist() == [3, 0, 0]  # ties (0.0) come back id-ascending
    assert I[1].tolist() == [1, 0, 2] and D[1].tolist() == [2, 1, 0]
    D, I = oracle.flat_search(xb, xq, 2, oracle.METRIC_L2)
    assert I[0].tolist() == [2, 0] and np.allclose(D[0], [4, 10])  # |q-e2|^2 = 4, |q-e0|^2 = 9+1


def test_k_larger_than_ntotal_pads_like_faiss():
    xb = np.array([[1, 0], [0, 1], [1, 1]], np.float32)
    D, I = oracle.flat_search(xb, np.array([[1, 0]], np.float32), 5)
    assert I.tolist() == [[0, 2, 1, -1, -1]]
    assert D[0, 3] == -FLT_MAX and D[0, 4] == -FLT_MAX
    D, I = oracle.flat_search(xb, np.array([[1, 0]], np.float32), 5, oracle.METRIC_L2)
    assert I.tolist() == [[0, 1, 2, -1, -1]] or I.tolist() == [[0, 2, 1, -1, -1]]
    assert D[0, 0] == 0 and D[0, 3] == FLT_MAX


def test_empty_shapes():
    xb = np.zeros((0, 4), np.float32)
    D, I = oracle.flat_search(xb, np.ones((2, 4), np.float32), 3)
    assert (I == -1).all() and (D == -FLT_MAX).all()
    D, I = oracle.flat_search(np.ones((5, 4), np.float32), np.zeros((0, 4), np.float32), 3)
    assert D.shape == (0, 3) and I.shape == (0, 3)
    D, I = oracle.flat_search(np.ones((5, 4), np.float32), np.ones((2, 4), np.float32), 0)
    assert D.shape == (2, 0)
    with pytest.raises(ValueError):
        oracle.flat_search(np.ones((5, 4), np.float32), np.ones((2, 3), np.float32), 1)


def test_duplicate_rows_tie_order_is_id_ascending():
    base = synth.corpus(20, 16, seed=1)
    xb = np.concatenate([base, base])
    D, I = oracle.flat_search(xb, base[:5], 4)
    for q in range(5):
        assert I[q, 0] == q and I[q, 1] == q + 20 and D[q, 0] == D[q, 1]


def test_fp64_and_noncontiguous_inputs_are_cast_like_the_faiss_wrapper():
    xb = synth.corpus(300, 24, seed=2)
    xq, _ = synth.queries(xb, 11)
    D0, I0 = oracle.flat_search(xb, xq, 5)
    D1, I1 = oracle.flat_search(xb.astype(np.float64), xq.astype(np.float64), 5)
    assert np.array_equal(I0, I1) and np.array_equal(D0, D1)
    D2, I2 = oracle.flat_search(np.asfortranarray(xb), xq[:, ::1], 5)
    assert np.array_equal(I0, I2)


@pytest.mark.parametrize("metric", [oracle.METRIC_INNER_PRODUCT, oracle.METRIC_L2])
@pytest.mark.parametrize("nq", [7, 64])  # below / above faiss's BLAS threshold of 20 queries
def test_three_implementations_agree(metric, nq):
    xb = synth.corpus(5000, 48, seed=3) * 1.3
    xq, _ = synth.queries(xb, nq, seed=5)
    Dn, In = oracle.flat_search(xb, xq, 9, metric, use_c=False)
    Dc, Ic = oracle.flat_search(xb, xq, 9, metric, use_c=True)
    assert np.array_equal(In, Ic) and np.array_equal(Dn, Dc)
    De, Ie = oracle.flat_search_exact64(xb, xq, 9, metric)
    err, hard, recall = synth.compare_topk(De, Ie, Dn, In)
    assert err < 2e-6 and hard == 0 and recall == 1.0
    Dk, Ik = _finish(cbind.flat_search_naive(xb, xq, 9, metric), metric, None)
    err, hard, recall = synth.compare_topk(De, Ie, Dk, Ik)
    assert err < 2e-6 and hard == 0 and recall == 1.0


def test_ids_subset_equals_search_on_gathered_rows():
    xb = synth.corpus(800, 32, seed=4)
    xq, _ = synth.queries(xb, 25)
    ids = np.random.default_rng(0).choice(800, 123, replace=False)
    D, I = oracle.flat_search(xb, xq, 6, ids=ids)
    Dg, Ig = oracle.flat_search(xb[ids], xq, 6)
    assert np.array_equal(D, Dg) and np.array_equal(I, ids[Ig])
    assert set(I.ravel()) <= set(ids.tolist())


def test_permutation_invariance_of_the_result_set():
    xb = synth.corpus(600, 32, seed=6)
    xq, _ = synth.queries(xb, 20)
    perm = np.random.default_rng(1).permutation(600)
    D, I = oracle.flat_search(xb, xq, 5)
    Dp, Ip = oracle.flat_search(xb[perm], xq, 5)
    assert np.allclose(D, Dp, atol=1e-6)
    assert np.array_equal(np.sort(I, 1), np.sort(perm[Ip], 1))


def test_key_order_is_score_then_id():
    s = np.array([-np.inf, -3.5, -0.0, 0.0, 1e-30, 2.0, np.inf], np.float32)
    o = _ord32(s)
    assert (np.diff(o.astype(np.int64)) >= 0).all() and o[2] == o[3]
    assert np.array_equal(_unord32(o)[[0, 1, 4, 5, 6]], s[[0, 1, 4, 5, 6]])
    k = oracle.pack_keys(np.array([1.0, 1.0, 2.0], np.float32), np.array([7, 3, 9]))
    assert k[2] > k[1] > k[0] > 0  # higher score first, then lower id
    better, ids, empty = oracle.unpack_keys(np.array([0, k[1]], np.uint64))
    assert ids.tolist() == [-1, 3] and empty.tolist() == [True, False] and better[1] == 1.0


def test_mt19937_known_answers():
    """std::mt19937 known-answer: the 10000th draw of the default-seeded (5489) engine is 4123659995 ([rand.predef]);
    seed 1234 starts 822569775, 2137449171, 2671936806 (checked against libstdc++ in this image)."""
    assert int(cbind.mt19937_raw(5489, 10000)[-1]) == 4123659995
    assert cbind.mt19937_raw(1234, 3).tolist() == [822569775, 2137449171, 2671936806]
    assert oracle.kmeans._mt19937_raw(5489, 10000)[-1] == 4123659995
    p = oracle.rand_perm(1000, 1234, use_c=True)
    assert np.array_equal(p, oracle.rand_perm(1000, 1234, use_c=False))
    assert sorted(p.tolist()) == list(range(1000))


def test_kmeans_restatement_properties():
    rng = np.random.default_rng(0)
    k, d = 6, 16
    centers = rng.standard_normal((k, d)).astype(np.float32) * 5
    lab = rng.integers(0, k, 1500)
    x = (centers[lab] + 0.3 * rng.standard_normal((1500, d))).astype(np.float32)
    r = oracle.kmeans_faiss(x, k, niter=10)
    r2 = oracle.kmeans_faiss(x, k, niter=10, use_c=False)
    assert np.array_equal(r.assign, r2.assign) and np.array_equal(r.centroids, r2.centroids)
    assert (np.diff(r.obj[:5]) <= 1e-3 * r.obj[0]).all()  # Lloyd objective is non-increasing (up to split noise)
    # every final assignment is the nearest centroid
    d2 = ((x[:, None, :].astype(np.float64) - r.centroids[None].astype(np.float64)) ** 2).sum(-1)
    assert (d2.argmin(1) == r.assign).mean() > 0.999
    # initial centroids are x[rand_perm(n, seed + 1)[:k]] (Appendix A.4): niter = 0 exposes them
    r0 = oracle.kmeans_faiss(x, k, niter=0)
    assert np.array_equal(r0.centroids, x[oracle.rand_perm(1500, 1235)[:k]])
    # subsampling to k * max_points_per_centroid rows via rand_perm(n, seed)
    rs = oracle.kmeans_faiss(x, 3, niter=2, max_points_per_centroid=100)
    assert np.array_equal(rs.train_ids, oracle.rand_perm(1500, 1234)[:300]) and len(rs.assign) == 1500
    # n == k copies the points
    rk = oracle.kmeans_faiss(x[:k], k, niter=5)
    assert np.array_equal(rk.centroids, x[:k]) and sorted(rk.assign.tolist()) == list(range(k))
    with pytest.raises(ValueError):
        oracle.kmeans_faiss(x[:3], 5)


def test_kmeans_empty_cluster_split():
    rng = np.random.default_rng(3)
    xd = np.repeat(rng.standard_normal((3, 8)).astype(np.float32), 20, axis=0)  # only 3 distinct points, k = 5
    a = oracle.kmeans_faiss(xd, 5, niter=4, use_c=True)
    b = oracle.kmeans_faiss(xd, 5, niter=4, use_c=False)
    assert a.nsplit.tolist() == b.nsplit.tolist() and a.nsplit[0] >= 2
    assert np.array_equal(a.centroids, b.centroids) and np.array_equal(a.assign, b.assign)


def test_dedup_restatement():
    e = np.eye(4, dtype=np.float32)
    X = np.stack([e[0], 0.99 * e[0] + 0.1 * e[1], e[0], e[2], 0.999 * e[2] + 0.02 * e[3]])
    X /= np.linalg.norm(X, axis=1, keepdims=True)
    vals = ["a", "b", "a", "c", "d"]
    i, j, s = oracle.range_self_join(X, 0.9)
    assert list(zip(i.tolist(), j.tolist())) == [(0, 1), (0, 2), (1, 0), (1, 2), (2, 0), (2, 1), (3, 4), (4, 3)]
    assert (s > 0.9).all()
    # rows 0 and 2 hold the same VALUE (kept together, sem_dedup.py:47,54); "b" and "d" are dropped
    assert oracle.dedup_keep_mask(vals, X, 0.9).tolist() == [True, False, True, True, False]
    assert oracle.dedup_keep_mask(vals, X, 0.9999).tolist() == [True] * 5
    lab = oracle.dedup_components(6, np.array([0, 4, 5]), np.array([3, 5, 2]))
    assert lab.tolist() == [0, 1, 2, 0, 2, 2]


@pytest.mark.parametrize("which", ["c", "torch"])
def test_cpu_timing_comparators_agree_with_the_oracle(which):
    """bench.py's cpu_baseline times oracle/blas_twin (C + OpenMP AVX-512 twin, torch-CPU fallback): both must return
    the oracle's answers (ids outside near-ties, scores to 1e-5 / 4e-5), padding included."""
    import synth
    from oracle import blas_twin

    fn = blas_twin.flat_search_c if which == "c" else blas_twin.flat_search_blas
    if which == "c" and not blas_twin.c_available():
        pytest.skip("C comparator not built")
    xb = synth.corpus(7001, 200, seed=1)
    xq, _ = synth.queries(xb, 333, seed=2)
    for metric, scale, atol in ((0, 1.0, 1e-5), (1, 1.4, 4e-5)):
        D, I, threads = fn(xb * scale, xq, 9, metric)
        Dr, Ir = oracle.flat_search(xb * scale, xq, 9, metric)
        err, hard, recall = synth.compare_topk(Dr, Ir, D, I, atol=atol)
        assert threads >= 1 and err <= atol and hard == 0 and recall >= 0.9999
    D, I, _ = fn(xb[:5], xq[:3], 8)
    assert (I[:, 5:] == -1).all() and (np.sort(I[:, :5], axis=1) == np.arange(5)).all()
    assert np.all(D[:, 5:] == -np.float32(3.4028234663852886e38))


def test_search_agrees_with_an_independent_brute_force_library():
    """faiss is not installable here, so the oracle cannot be run against it ("parity unpinned").  As an
    independent third-party check, scikit-learn's brute-force NearestNeighbors (its own pairwise-distance code) must
    return the same neighbours wherever adjacent distances are not within rounding of each other, for L2 and for
    inner product on unit vectors (cosine distance = 1 - ip)."""
    from sklearn.neighbors import NearestNeighbors

    for seed, (n, d, nq, k) in enumerate([(2000, 48, 64, 10), (5000, 96, 40, 5), (1500, 384, 32, 7)]):
        xb = synth.corpus(n, d, seed=seed + 10)
        xq, _ = synth.queries(xb, nq, seed=seed + 20)
        for metric, sk_metric in ((oracle.METRIC_L2, "euclidean"), (oracle.METRIC_INNER_PRODUCT, "cosine")):
            D, I = oracle.flat_search(xb, xq, k, metric)
            nn = NearestNeighbors(n_neighbors=k, algorithm="brute", metric=sk_metric).fit(xb.astype(np.float64))
            sd, si = nn.kneighbors(xq.astype(np.float64))
            ref = sd ** 2 if metric == oracle.METRIC_L2 else 1.0 - sd
            assert np.allclose(D, ref, atol=2e-5), (metric, np.abs(D - ref).max())
            # ids may only differ inside groups of near-equal scores
            for q in range(nq):
                if np.array_equal(I[q], si[q]):
                    continue
                bad = I[q] != si[q]
                gaps = np.abs(ref[q][bad][:, None] - ref[q][None, :])
                assert (np.sort(gaps, axis=1)[:, 1] < 4e-5).all(), (metric, q, I[q], si[q])


def test_lloyd_iterations_agree_with_an_independent_kmeans():
    """Same initial centroids (x[rand_perm(n, seed + 1)[:k]]), same number of Lloyd iterations, no subsampling and no empty
    cluster: scikit-learn's KMeans (its own E/M steps) must end at the oracle's centroids and assignment."""
    from sklearn.cluster import KMeans

    rng = np.random.default_rng(5)
    k, d, n, niter = 8, 16, 1600, 12
    centers = rng.standard_normal((k, d)).astype(np.float32) * 3
    x = (centers[rng.integers(0, k, n)] + rng.standard_normal((n, d)).astype(np.float32)).astype(np.float32)
    res = oracle.kmeans_faiss(x, k, niter=niter, seed=1234)
    assert res.nsplit.sum() == 0
    init = x[oracle.rand_perm(n, 1235)[:k]]
    sk = KMeans(n_clusters=k, init=init.astype(np.float64), n_init=1, max_iter=niter, tol=0.0, algorithm="lloyd")
    sk.fit(x.astype(np.float64))
    assert np.allclose(res.centroids, sk.cluster_centers_, atol=1e-4), np.abs(res.centroids - sk.cluster_centers_).max()
    assert (res.assign == sk.labels_).mean() > 0.999  # points within rounding of a cell border may differ
    # faiss's objective of the LAST iteration is measured against the centroids BEFORE that iteration's update
    assert res.obj[-1] >= sk.inertia_ * (1 - 1e-5)


def test_kmeans_trace_and_flipped_rows():
    """The per-iteration record a step-by-step parity check feeds to the device (tests/test_gpu_kmeans.py, bench.py) is
    self-consistent, and flipped_rows separates near-ties from real disagreements."""
    rng = np.random.default_rng(12)
    k, d, n = 24, 16, 4000
    c = rng.standard_normal((k, d)).astype(np.float32) * 3
    x = (c[rng.integers(0, 6, n)] + 0.2 * rng.standard_normal((n, d))).astype(np.float32)
    x[: n // 2] = x[n // 2: n // 2 + 40].repeat(n // 80, axis=0)[: n // 2]  # duplicate rows: duplicate initial centroids run empty
    trace = []
    r = oracle.kmeans_faiss(x, k, niter=5, trace=trace)
    assert len(trace) == 5 and r.nsplit.sum() > 0
    for it, rec in enumerate(trace):
        D, I = oracle.flat_search(rec["centroids"], x, 1, oracle.METRIC_L2)
        assert np.array_equal(I[:, 0], rec["assign"]) and np.array_equal(D[:, 0], rec["dist"])
        assert np.array_equal(np.bincount(rec["assign"], minlength=k).astype(np.float32), rec["hassign"])
        assert rec["hassign_after"].sum() == n and (rec["hassign"] == 0).sum() == r.nsplit[it]
        if it + 1 < len(trace):
            assert np.array_equal(rec["next"], trace[it + 1]["centroids"])
    assert np.array_equal(trace[-1]["next"], r.centroids)
    # pair_distances restates the search's expression
    rec = trace[2]
    rows = np.arange(0, n, 7)
    pd = oracle.pair_distances(x, rec["centroids"], rows, rec["assign"][rows])
    big = float((x ** 2).sum(1).max() + (rec["centroids"] ** 2).sum(1).max())  # the expression's largest term
    assert np.abs(pd - rec["dist"][rows]).max() <= 4 * np.finfo(np.float32).eps * big
    # flipped_rows: an exact tie (two identical centroids) is a near-tie, a real change of cluster is not
    cent = rec["centroids"].copy()
    a = rec["assign"].copy()
    cent[23] = cent[a[0]]
    b = a.copy()
    b[0] = 23
    fl = oracle.flipped_rows(x, cent, a, b)
    assert fl["rows"].tolist() == [0] and fl["all_near_ties"] and fl["gaps"][0] == 0
    far = int(np.argmax(((x[1][None, :] - cent) ** 2).sum(1)))
    b[1] = far
    fl = oracle.flipped_rows(x, cent, a, b)
    assert fl["rows"].tolist() == [0, 1] and fl["near_tie"].tolist() == [True, False] and not fl["all_near_ties"]
