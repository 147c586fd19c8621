"""Host logic of lotus_amd.cluster (faiss-parity k-means driver + the lotus.utils.cluster replacement) on the CPU
with the oracle-backed test double; the device kernels it drives are covered by tests/test_gpu_kmeans.py."""
import numpy as np
import pandas as pd
import pytest

import oracle
import ref_harness
from oracle_backend import OracleBackend, _emulate_storage


def blobs(n=1500, k=6, d=16, seed=0):
    rng = np.random.default_rng(seed)
    c = rng.standard_normal((k, d)).astype(np.float32) * 4
    return (c[rng.integers(0, k, n)] + 0.3 * rng.standard_normal((n, d))).astype(np.float32)


def test_kmeans_driver_equals_the_oracle_restatement():
    from lotus_amd.cluster import kmeans

    x = blobs()
    x = _emulate_storage(x, 1)  # values the device would hold, so both sides see identical inputs
    r = kmeans(x, 6, niter=7, backend=OracleBackend())
    o = oracle.kmeans_faiss(x, 6, niter=7)
    assert np.array_equal(r.assign, o.assign) and np.allclose(r.centroids, o.centroids, atol=1e-6)
    assert np.allclose(r.obj, o.obj, rtol=1e-6) and np.array_equal(r.train_ids, o.train_ids)


def test_subsample_n_equals_k_and_errors():
    from lotus_amd.cluster import kmeans

    x = _emulate_storage(blobs(900, 3, 8, seed=2), 1)
    r = kmeans(x, 3, niter=3, max_points_per_centroid=100, backend=OracleBackend())
    o = oracle.kmeans_faiss(x, 3, niter=3, max_points_per_centroid=100)
    assert len(r.train_ids) == 300 and np.array_equal(r.train_ids, o.train_ids) and np.array_equal(r.assign, o.assign)
    full = kmeans(x, 3, niter=3, max_points_per_centroid=None, backend=OracleBackend())
    assert len(full.train_ids) == 900
    rk = kmeans(x[:5], 5, niter=4, backend=OracleBackend())
    assert np.array_equal(rk.centroids, x[:5])
    with pytest.raises(ValueError):
        kmeans(x[:3], 5, backend=OracleBackend())


def test_empty_cluster_split_path():
    from lotus_amd.cluster import kmeans

    xd = np.repeat(np.random.default_rng(3).standard_normal((3, 8)).astype(np.float32), 20, axis=0)
    xd = _emulate_storage(xd, 1)
    r = kmeans(xd, 5, niter=4, backend=OracleBackend())
    o = oracle.kmeans_faiss(xd, 5, niter=4)
    assert r.nsplit.tolist() == o.nsplit.tolist() and r.nsplit[0] >= 2
    assert np.allclose(r.centroids, o.centroids, atol=1e-6) and np.array_equal(r.assign, o.assign)


@pytest.mark.skipif(not ref_harness.available(), reason="reference checkout not present")
def test_installed_cluster_matches_the_reference_function(tmp_path):
    """sem_cluster_by through the REAL accessor: reference lotus.utils.cluster (faiss shim) vs our replacement."""
    lotus = ref_harness.import_lotus()
    from lotus.models.rm import RM
    from lotus.vector_store.faiss_vs import FaissVS

    import fake_rm
    from lotus_amd import HipVS, cluster as lcluster

    words = sum(fake_rm.TOPICS.values(), [])
    rng = np.random.default_rng(1)
    texts = [" ".join(rng.choice(words, 3)) for _ in range(90)]

    def run(d):
        df = pd.DataFrame({"T": texts}).sem_index("T", d)
        return df.sem_cluster_by("T", 4, niter=6)

    lotus.settings.configure(rm=fake_rm.make_rm(RM), vs=FaissVS())
    a = run(str(tmp_path / "f"))
    lotus.settings.configure(rm=fake_rm.make_rm(RM), vs=HipVS(backend=OracleBackend()))
    lcluster.install()
    try:
        assert lotus.utils.cluster is lcluster.cluster
        b = run(str(tmp_path / "h"))
        df = pd.DataFrame({"T": texts}).sem_index("T", str(tmp_path / "h2"))
        with pytest.raises(ValueError, match="Number of centroids"):
            df.sem_cluster_by("T", 91)
        with pytest.raises(ValueError, match="not found"):
            lcluster.cluster("missing", 2)(df)
    finally:
        lcluster.uninstall()
    assert lotus.utils.cluster is not lcluster.cluster
    assert a["cluster_id"].tolist() == b["cluster_id"].tolist() and len(set(a["cluster_id"])) == 4


def test_step_by_step_harness_on_the_cpu_double():
    """tests/km_steps.py (the teacher-forced and free-run k-means parity checks the GPU suite runs at configs[4]'s shape)
    over the oracle-backed test double on a small blob set with empty-cluster splits: the harness itself is sound."""
    import km_steps
    from oracle_backend import OracleBackend

    rng = np.random.default_rng(5)
    K, d, n = 48, 24, 20_000
    cen = rng.standard_normal((12, d)).astype(np.float32)
    x = (cen[rng.integers(0, 12, n)] + 0.15 * rng.standard_normal((n, d))).astype(np.float16)
    x[: n // 4] = x[n // 2: n // 2 + 10].repeat(n // 40, axis=0)[: n // 4]  # duplicates: duplicate initial centroids run empty
    c = km_steps.reference(x, K, 5)
    assert len(c["ref"].train_ids) == K * 256 and c["ref"].nsplit.sum() > 0
    be = OracleBackend()
    # the double keeps the centroids as fp16 hi|lo pairs like the device (~22 significant bits): a handful of near-tie flips
    # against the float32 oracle are legitimate - the harness has checked that each one IS a near-tie
    assert km_steps.teacher_forced(be, c) <= 2
    rep = km_steps.free_run(be, c)
    assert rep["iterations_compared"] == 5 and rep["all_flips_are_near_ties"]


def test_row_ranges_of_an_iteration_cover_the_rows_in_order():
    """`cluster.range_cuts`: the ranges an exhaustive k-means iteration hands the rows over in (sums of one range under the
    search of the next) - consecutive, complete, cut on multiples of 4 096 rows, in the requested proportions."""
    from lotus_amd import cluster

    for n in (65536 * 4, 2_097_152, 10_000_000, 3_000_001):
        for fracs in ((0.25,) * 4, cluster.PARTS_DEFAULT, (3, 3, 2.5, 1.5), (1.0,), (0.5, 0.5)):
            cuts = cluster.range_cuts(n, fracs)
            assert cuts[0] == 0 and cuts[-1] == n and len(cuts) == len(fracs) + 1
            assert all(b >= a for a, b in zip(cuts, cuts[1:])) and all(c % 4096 == 0 for c in cuts[1:-1])
            total = float(sum(fracs))
            for i, f in enumerate(fracs[:-1]):
                assert abs((cuts[i + 1] - cuts[i]) - n * f / total) <= 2 * 4096
    assert abs(sum(cluster.PARTS_DEFAULT) - 1.0) < 1e-9 and min(cluster.PARTS_DEFAULT) == cluster.PARTS_DEFAULT[-1]
