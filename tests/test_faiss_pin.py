"""The pin the oracle is waiting for ("parity unpinned" - faiss-cpu 1.13.0, the reference's dependency,
`uv.lock:573-574`, is neither vendored nor installable in this image).  Every test here is skipped until `import faiss` works;
the day a wheel exists they hold the CPU restatement (`oracle/`), the on-disk interop (`lotus_amd/faiss_io.py`) and the faiss-shaped
test double (`tests/fake_faiss.py`) against the real library on the committed golden inputs - no GPU, no other change needed.
What each one pins, by reference line:
  * flat search  - `FaissVS.__call__` -> `index.search` (`lotus/vector_store/faiss_vs.py:67,75`), both metrics, padding, subsets
  * k-means      - `faiss.Kmeans(d, k, niter=, verbose=).train` + `index.search(x, 1)` (`lotus/utils.py:61-65`)
  * index files  - `faiss.write_index` / `read_index` (`faiss_vs.py:30,34`)
`tests/ref_harness.py` prefers the real module for the reference's own operator runs, and `bench.py`'s `cpu_baseline` times the
real `index_factory('Flat').search` (kind "reference") under the same condition."""
import glob
import os

import numpy as np
import pytest

import oracle
import synth

faiss = pytest.importorskip("faiss", reason="faiss-cpu is not installable in this image (SURVEY.md 8(c)); these tests pin the oracle the day it is")
HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.join(HERE, "golden")


def _faiss_search(xb, xq, k, metric):
    index = faiss.index_factory(int(xb.shape[1]), "Flat", faiss.METRIC_INNER_PRODUCT if metric == 0 else faiss.METRIC_L2)
    index.add(np.ascontiguousarray(xb, dtype=np.float32))
    return index.search(np.ascontiguousarray(xq, dtype=np.float32), int(k))


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLDEN, "flat_*.npz"))))
def test_oracle_flat_search_equals_faiss_on_the_golden_inputs(path):
    g = np.load(path)
    xb, xq, k, metric, ids = g["xb"], g["xq"], int(g["k"]), int(g["metric"]), g["ids"]
    if ids.size:  # the ids branch of FaissVS.__call__ (faiss_vs.py:57-72): a temporary index over the subset, remapped
        Df, If = _faiss_search(xb[ids], xq, k, metric)
        If = np.where(If >= 0, ids[np.maximum(If, 0)], -1)
    else:
        Df, If = _faiss_search(xb, xq, k, metric)
    Do, Io = oracle.flat_search(xb, xq, k, metric, ids=ids if ids.size else None)
    err, hard, recall = synth.compare_topk(Df, If, Do, Io, atol=1e-6)
    assert err <= 1e-6 and hard == 0 and recall == 1.0
    assert np.array_equal(If < 0, Io < 0)                       # same padding slots ...
    assert np.array_equal(Df[If < 0], Do[Io < 0])               # ... with the same -FLT_MAX / +FLT_MAX
    # and the committed expectations (what the GPU tests compare with) are faiss's answers too
    err, hard, recall = synth.compare_topk(Df, If, g["D"], g["I"], atol=1e-6)
    assert err <= 1e-6 and hard == 0 and recall == 1.0


@pytest.mark.parametrize("nq", [1, 19, 20, 21, 300])  # faiss switches from direct distances to the BLAS path at 20 queries
@pytest.mark.parametrize("metric", [0, 1])
def test_oracle_flat_search_equals_faiss_around_the_blas_threshold(nq, metric):
    xb = synth.corpus(5000, 96, seed=5) * (1.4 if metric else 1.0)
    xq, _ = synth.queries(xb, nq, seed=6)
    Df, If = _faiss_search(xb, xq, 10, metric)
    Do, Io = oracle.flat_search(xb, xq, 10, metric)
    err, hard, recall = synth.compare_topk(Df, If, Do, Io, atol=1e-6 if metric == 0 else 4e-6)
    assert hard == 0 and recall == 1.0 and err <= (1e-6 if metric == 0 else 4e-6)


def test_oracle_kmeans_equals_faiss_kmeans():
    g = np.load(os.path.join(GOLDEN, "kmeans_blobs.npz"))
    x, k, niter = g["x"], int(g["k"]), int(g["niter"])
    km = faiss.Kmeans(int(x.shape[1]), k, niter=niter, verbose=False)   # lotus/utils.py:61
    km.max_points_per_centroid = int(g["mppc"])
    km.train(np.ascontiguousarray(x, dtype=np.float32))                  # :62
    _, assign = km.index.search(np.ascontiguousarray(x, dtype=np.float32), 1)  # :65
    res = oracle.kmeans_faiss(x, k, niter=niter, max_points_per_centroid=int(g["mppc"]))
    assert np.allclose(km.centroids.reshape(k, -1), res.centroids, rtol=0, atol=1e-5)
    assert (assign[:, 0] == res.assign).mean() >= 1 - 1e-4
    fobj = np.array([km.obj[i] for i in range(len(km.obj))], np.float64) if hasattr(km, "obj") else None
    if fobj is not None and len(fobj) == niter:
        assert np.allclose(fobj, res.obj, rtol=1e-5)
    # the committed fixture (what the GPU k-means tests compare with) is faiss's result as well
    assert np.allclose(km.centroids.reshape(k, -1), g["centroids"], rtol=0, atol=1e-5)
    assert (assign[:, 0] == g["assign"]).mean() >= 1 - 1e-4


def test_mt19937_permutation_is_the_one_faiss_draws():
    """faiss::rand_perm / RandomGenerator (std::mt19937) - the subsample and the initial centroids of Kmeans.train."""
    for n, seed in ((10, 1234), (1000, 1235), (4097, 7)):
        fp = faiss.rand_perm(n, seed) if hasattr(faiss, "rand_perm") else None
        if fp is None:
            pytest.skip("this faiss build does not export rand_perm")
        assert np.array_equal(np.asarray(fp, np.int64), oracle.rand_perm(n, seed))


@pytest.mark.parametrize("metric", [0, 1])
def test_index_files_are_interchangeable_with_faiss(tmp_path, metric):
    from lotus_amd import faiss_io

    x = synth.corpus(37, 24, seed=9)
    ours, theirs = str(tmp_path / "ours.index"), str(tmp_path / "theirs.index")
    faiss_io.write_index_flat(ours, x, metric)
    index = faiss.index_factory(24, "Flat", faiss.METRIC_INNER_PRODUCT if metric == 0 else faiss.METRIC_L2)
    index.add(x)
    faiss.write_index(index, theirs)                                   # faiss_vs.py:30
    assert open(ours, "rb").read() == open(theirs, "rb").read()        # byte for byte
    back = faiss.read_index(ours)                                      # faiss_vs.py:34 reads what we wrote
    assert back.ntotal == 37 and back.d == 24
    xr, mr = faiss_io.read_index_flat(theirs)                          # ... and we read what faiss wrote
    assert mr == metric and np.array_equal(xr, x)
    # the hand-packed byte fixtures the CPU suite pins the layout with are real faiss files
    for path in glob.glob(os.path.join(GOLDEN, "faiss_flat_*.index")):
        idx = faiss.read_index(path)
        xr, _ = faiss_io.read_index_flat(path)
        assert idx.ntotal == xr.shape[0] and idx.d == xr.shape[1]
        assert np.array_equal(faiss.vector_to_array(idx.codes).view(np.float32).reshape(xr.shape), xr)


def test_the_test_double_behaves_like_the_real_module():
    """tests/fake_faiss.py stands in for faiss when the reference's own FaissVS / lotus.utils.cluster run here: same answers."""
    import fake_faiss

    xb = synth.corpus(800, 48, seed=2)
    xq, _ = synth.queries(xb, 33, seed=3)
    for mod_metric, metric in ((("METRIC_INNER_PRODUCT"), 0), (("METRIC_L2"), 1)):
        a = faiss.index_factory(48, "Flat", getattr(faiss, mod_metric))
        b = fake_faiss.index_factory(48, "Flat", getattr(fake_faiss, mod_metric))
        a.add(xb)
        b.add(xb)
        Da, Ia = a.search(xq, 7)
        Db, Ib = b.search(xq, 7)
        err, hard, recall = synth.compare_topk(Da, Ia, Db, Ib, atol=4e-6)
        assert hard == 0 and recall == 1.0 and err <= 4e-6
