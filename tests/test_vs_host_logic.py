"""Host-side logic of HipVS (argument handling, padding, ids subset mapping, residency, persistence) exercised on
the CPU with the oracle-backed test double.  The compute path proper is covered by the `-m gpu` tests."""
import os
import pickle

import numpy as np
import pytest

import oracle
import synth
from lotus_amd import HipVS, METRIC_INNER_PRODUCT, METRIC_L2, RMOutput, VS, _capi, faiss_io
from lotus_amd.vs import FLT_MAX
from oracle_backend import OracleBackend, _emulate_storage


def make_vs(**kw):
    return HipVS(backend=OracleBackend(), **kw)


@pytest.fixture
def indexed(tmp_path):
    xb = synth.corpus(400, 48, seed=11)
    vs = make_vs()
    d = str(tmp_path / "idx")
    vs.index(["doc"] * 400, xb, d)
    return vs, xb, d


def test_is_a_vs_subclass_and_sets_index_dir(indexed):
    vs, xb, d = indexed
    assert isinstance(vs, VS) and vs.index_dir == d
    assert sorted(os.listdir(d)) == ["index", "rows.json", "vecs"]  # the reference's two files + the row-store description
    with open(os.path.join(d, "vecs"), "rb") as fp:
        assert np.array_equal(pickle.load(fp), xb)  # same pickle the reference writes (faiss_vs.py:27-29)
    x, metric = faiss_io.read_index_flat(os.path.join(d, "index"))
    assert metric == 0 and np.array_equal(x, xb)


def test_search_matches_oracle_and_result_types(indexed):
    vs, xb, _ = indexed
    xq, _ = synth.queries(xb, 9)
    out = vs(xq, 5)
    assert isinstance(out, RMOutput)
    assert out.distances.dtype == np.float32 and out.indices.dtype == np.int64
    D, I = oracle.flat_search(_emulate_storage(xb, _capi.PACK_SPLIT), _emulate_storage(xq, _capi.PACK_SPLIT), 5)
    assert np.array_equal(out.indices, I) and np.allclose(out.distances, D, atol=1e-6)
    assert (np.diff(out.distances, axis=1) <= 0).all()  # best first


def test_not_loaded_raises_value_error():
    with pytest.raises(ValueError, match="Index not loaded"):
        make_vs()(np.zeros((1, 4), np.float32), 1)


def test_k_zero_and_empty_queries(indexed):
    vs, xb, _ = indexed
    out = vs(xb[:3], 0)
    assert out.distances.shape == (3, 0) and out.indices.shape == (3, 0)
    out = vs(np.zeros((0, 48), np.float32), 4)
    assert out.distances.shape == (0, 4)


def test_k_beyond_ntotal_is_padded(tmp_path):
    xb = synth.corpus(7, 16, seed=1)
    vs = make_vs()
    vs.index(None, xb, str(tmp_path / "i"))
    out = vs(xb[:2], 8)  # sem_search doubles K past ntotal (sem_search.py:136-138)
    assert (out.indices[:, 7] == -1).all() and (out.distances[:, 7] == -FLT_MAX).all()
    assert sorted(out.indices[0, :7].tolist()) == list(range(7))
    vs2 = make_vs(metric=METRIC_L2)
    vs2.index(None, xb, str(tmp_path / "j"))
    out = vs2(xb[:2], 8)
    assert (out.indices[:, 7] == -1).all() and (out.distances[:, 7] == FLT_MAX).all()
    assert out.indices[0, 0] == 0 and out.distances[0, 0] <= 1e-6  # nearest by squared L2 is itself


def test_dimension_mismatch_and_bad_args(indexed):
    vs, xb, _ = indexed
    with pytest.raises(ValueError):
        vs(np.zeros((2, 47), np.float32), 3)
    with pytest.raises(ValueError):
        vs(xb[:2], -1)
    with pytest.raises(IndexError):
        vs(xb[:2], 3, ids=[0, 400])
    with pytest.raises(ValueError):
        HipVS(metric=7)
    with pytest.raises(ValueError):
        HipVS(storage="int8")


def test_one_dimensional_query_and_fp64(indexed):
    vs, xb, _ = indexed
    a = vs(xb[5], 3)
    b = vs(xb[5:6].astype(np.float64), 3)  # LiteLLM embeddings are float64 (litellm_rm.py:69)
    assert a.indices.shape == (1, 3) and np.array_equal(a.indices, b.indices) and a.indices[0, 0] == 5


def test_ids_full_range_skips_the_gather(indexed):
    vs, xb, _ = indexed
    xq, _ = synth.queries(xb, 6)
    ref = vs(xq, 4)
    vs.backend.calls.clear()
    out = vs(xq, 4, ids=list(range(400)))  # what sem_sim_join always passes (sem_sim_join.py:132-134)
    assert np.array_equal(out.indices, ref.indices) and np.array_equal(out.distances, ref.distances)
    assert not any(c[0] == "gather" for c in vs.backend.calls)


def test_ids_strict_subset_returns_original_row_numbers(indexed):
    vs, xb, _ = indexed
    xq, _ = synth.queries(xb, 12)
    ids = np.random.default_rng(3).choice(400, 57, replace=False).tolist()
    out = vs(xq, 6, ids=ids)
    D, I = oracle.flat_search(_emulate_storage(xb, 1), _emulate_storage(xq, 1), 6, ids=ids)
    assert np.array_equal(out.indices, I) and np.allclose(out.distances, D, atol=1e-6)
    assert set(out.indices.ravel().tolist()) <= set(ids)
    assert any(c[0] == "gather" for c in vs.backend.calls)
    out = vs(xq, 80, ids=ids)  # K beyond the subset size pads
    assert (out.indices[:, 57:] == -1).all() and (out.indices[:, :57] >= 0).all()
    out = vs(xq, 3, ids=[])
    assert (out.indices == -1).all()


def test_get_vectors_from_index_accepts_lists_and_pandas_index(indexed):
    pd = pytest.importorskip("pandas")
    vs, xb, d = indexed
    assert np.array_equal(vs.get_vectors_from_index(d, [3, 1, 2]), xb[[3, 1, 2]])
    assert np.array_equal(vs.get_vectors_from_index(d, pd.RangeIndex(5)), xb[:5])  # sem_sim_join.py:115
    assert np.array_equal(vs.get_vectors_from_index(d, pd.Index([7, 9])), xb[[7, 9]])
    fresh = make_vs()  # not resident: falls back to the pickle like FaissVS (faiss_vs.py:38-41)
    assert np.array_equal(fresh.get_vectors_from_index(d, [0, 399]), xb[[0, 399]])
    assert vs.get_vectors_from_index(d, [0]).dtype == np.float32


def test_load_index_round_trip_and_residency(tmp_path):
    xa, xb = synth.corpus(50, 16, seed=1), synth.corpus(60, 16, seed=2)
    vs = make_vs(max_resident=2)
    da, db = str(tmp_path / "a"), str(tmp_path / "b")
    vs.index(None, xa, da)
    vs.index(None, xb, db)
    n_pack = sum(c[0] == "pack" for c in vs.backend.calls)
    vs.load_index(da)  # flip back (sem_sim_join.py:111-128): served from the resident copy, no re-pack
    assert vs.index_dir == da and sum(c[0] == "pack" for c in vs.backend.calls) == n_pack
    assert vs(xa[:1], 1).indices[0, 0] == 0
    fresh = make_vs()
    fresh.load_index(db)
    assert fresh.index_dir == db and fresh(xb[7:8], 1).indices[0, 0] == 7
    # an index directory written by stock LOTUS holds the same two files; a bare faiss file also loads
    os.remove(os.path.join(db, "vecs"))
    other = make_vs()
    other.load_index(db)
    assert other(xb[9:10], 1).indices[0, 0] == 9
    # eviction beyond max_resident
    vs.index(None, xa, str(tmp_path / "c"))
    assert len(vs._resident) == 2


def test_storage_modes_select_the_pack_layout(tmp_path):
    x32 = synth.corpus(30, 16, seed=4)
    for storage, dtype, mode in (("auto", np.float32, _capi.PACK_SPLIT), ("auto", np.float16, _capi.PACK_F16),
                                 ("auto", np.float64, _capi.PACK_SPLIT), ("fp16", np.float32, _capi.PACK_F16),
                                 ("fp32", np.float16, _capi.PACK_SPLIT)):
        vs = make_vs(storage=storage)
        vs.index(None, x32.astype(dtype), str(tmp_path / f"{storage}{np.dtype(dtype).name}"))
        assert vs.backend.calls[0] == ("pack", (30, 16), mode)


def test_k_equal_n_and_beyond_max_k(indexed, tmp_path):
    vs, xb, _ = indexed
    assert vs(xb[:1], 400).indices.shape == (1, 400)  # K = N callers (sem_dedup.py:45) at small N
    big = synth.corpus(2600, 8, seed=2)  # more rows than LVS_MAX_K: full ranking path
    v2 = make_vs()
    v2.index(None, big, str(tmp_path / "big"))
    out = v2(big[:3], 2600)
    D, I = oracle.flat_search(_emulate_storage(big, 1), _emulate_storage(big[:3], 1), 2600)
    assert np.array_equal(out.indices, I) and ("rank", 3, 2600) in v2.backend.calls
    sub = list(range(0, 2600, 1))[5:]
    assert v2(big[:2], 2595, ids=sub).indices.shape == (2, 2595)


def test_scores_matrix_for_cascade_callers(indexed):
    vs, xb, _ = indexed
    xq, _ = synth.queries(xb, 5)
    S = vs.scores(xq)
    ref = _emulate_storage(xq, 1) @ _emulate_storage(xb, 1).T
    assert S.shape == (5, 400) and S.dtype == np.float32 and np.allclose(S, ref, atol=1e-6)
    sub = [3, 7, 399]
    assert np.allclose(vs.scores(xq, ids=sub), ref[:, sub], atol=1e-6)


def test_normalize_option_is_cosine_similarity(tmp_path):
    """HipVS(normalize=True): rows and queries are L2-normalised while they are packed - inner product == cosine."""
    import synth
    from lotus_amd import HipVS
    from oracle_backend import OracleBackend

    xb = synth.corpus(300, 16, seed=3) * np.linspace(0.1, 9.0, 300, dtype=np.float32)[:, None]
    xq = synth.queries(synth.corpus(300, 16, seed=3), 7, seed=4)[0] * 5.0
    vs = HipVS(backend=OracleBackend(), normalize=True)
    vs.index(None, xb, str(tmp_path / "i"), persist=False)
    out = vs(xq, 4)
    cos = (xq / np.linalg.norm(xq, axis=1, keepdims=True)) @ (xb / np.linalg.norm(xb, axis=1, keepdims=True)).T
    ref = np.argsort(-cos, axis=1)[:, :4]
    assert np.array_equal(out.indices, ref) and np.allclose(out.distances, np.take_along_axis(cos, ref, 1), atol=1e-5)


def test_k_equals_n_ranking_goes_through_in_query_blocks(tmp_path, monkeypatch):
    """K = N callers beyond LVS_MAX_K: the score matrix is produced and ranked in blocks of <= 2^28 scores."""
    import synth
    from lotus_amd import HipVS, vs as vs_mod
    from oracle_backend import OracleBackend

    xb = synth.corpus(2100, 8, seed=5)
    vs = HipVS(backend=OracleBackend())
    vs.index(None, xb, str(tmp_path / "i"), persist=False)
    calls = []
    orig = vs._score_rows
    monkeypatch.setattr(vs, "_score_rows", lambda ent, q, sub, world, **kw: (calls.append(q.n), orig(ent, q, sub, world, **kw))[1])
    out = vs(xb[:50], 2100)
    assert calls == [50]
    monkeypatch.setattr(vs_mod, "_RANK_BLOCK_SCORES", 2100 * 16)  # 16 queries per block
    calls.clear()
    out2 = vs(xb[:50], 2100)
    assert calls == [16, 16, 16, 2] and np.array_equal(out2.indices, out.indices) and np.array_equal(out2.distances, out.distances)
    import oracle
    Dr, Ir = oracle.flat_search(xb, xb[:50], 2100)
    err, hard, recall = synth.compare_topk(Dr, Ir, out.distances, out.indices)
    assert err <= 1e-5 and hard == 0 and recall == 1.0


def test_query_validation_verdicts_on_every_result_path(tmp_path):
    """Queries are validated while they are packed: magnitudes that leave fp16's range under the index's scale are searched
    again with an exponent of their own (inner product), inf / NaN raise - from ``__call__`` and from ``scores()``."""
    rng = np.random.default_rng(12)
    xb = (0.05 * rng.standard_normal((400, 24))).astype(np.float32)
    xq = rng.standard_normal((6, 24)).astype(np.float32)
    xq[4] *= 1.0e7
    vs = HipVS(backend=OracleBackend())
    vs.index(None, xb, str(tmp_path / "i"), persist=False)
    out = vs(xq, 5)
    Dr, Ir = oracle.flat_search(xb, xq, 5, 0)
    assert np.array_equal(out.indices[:4], Ir[:4]) and np.array_equal(out.indices[4], Ir[4])
    assert np.allclose(out.distances, Dr, rtol=2e-5)
    S = vs.scores(xq)
    assert np.allclose(S, xq @ xb.T, rtol=1e-4, atol=1e-4 * np.abs(xq @ xb.T).max(axis=1, keepdims=True))
    bad = xq.copy()
    bad[1, 2] = np.inf
    for call in (lambda: vs(bad, 5), lambda: vs.scores(bad)):
        with pytest.raises(ValueError, match="inf or NaN"):
            call()
    l2 = HipVS(metric=METRIC_L2, backend=OracleBackend())
    l2.index(None, xb, str(tmp_path / "l2"), persist=False)
    with pytest.raises(ValueError, match="fp16's range"):  # squared L2 needs one scale on both sides: no retry
        l2(xq, 5)
