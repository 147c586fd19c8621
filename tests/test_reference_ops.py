"""Drop-in check against the reference's OWN operator code: the real `lotus` package is imported from
/root/reference (third-party modules stubbed, `faiss` mapped onto the oracle shim) and the scenarios of
.github/tests/rm_tests.py are replayed twice - once with the reference's unmodified FaissVS, once with HipVS -
and must give identical frames.  Here HipVS runs on the oracle-backed test double (no GPU in this container);
the same scenarios run on the real HIP path in tests/test_gpu_ops.py against the golden frames saved from here."""
import numpy as np
import pandas as pd
import pytest

import ref_harness

pytestmark = pytest.mark.skipif(not ref_harness.available(), reason="reference checkout not present")


@pytest.fixture(scope="module")
def env():
    lotus = ref_harness.import_lotus()
    from lotus.models.rm import RM
    from lotus.vector_store.faiss_vs import FaissVS

    import fake_rm
    from lotus_amd import HipVS
    from oracle_backend import OracleBackend

    return dict(lotus=lotus, FaissVS=FaissVS, HipVS=lambda **kw: HipVS(backend=OracleBackend(), **kw),
                rm=lambda dtype=np.float32: fake_rm.make_rm(RM, dtype))


def both(env, fn, tmp_path, **hip_kw):
    """Run scenario `fn(index_dir)` with the reference FaissVS and with HipVS; return both results."""
    lotus = env["lotus"]
    outs = []
    for name, vs in (("faiss", env["FaissVS"]()), ("hip", env["HipVS"](**hip_kw))):
        lotus.settings.configure(rm=env["rm"](), vs=vs)
        outs.append(fn(str(tmp_path / name)))
    return outs


def frames_equal(a, b):
    pd.testing.assert_frame_equal(a.reset_index(drop=True), b.reset_index(drop=True), check_dtype=False, atol=1e-5)
    assert a.index.tolist() == b.index.tolist()


COURSES = ["Probability and Random Processes", "Cooking", "Food Sciences", "Optimization Methods in Engineering"]


def test_search_rm_only(env, tmp_path):
    def run(d):
        df = pd.DataFrame({"Course Name": COURSES}).sem_index("Course Name", d)
        return df.sem_search("Course Name", "Optimization", K=1)

    a, b = both(env, run, tmp_path)
    assert a["Course Name"].tolist() == ["Optimization Methods in Engineering"]
    frames_equal(a, b)


def test_search_scores_and_k_equal_n(env, tmp_path):
    def run(d):
        df = pd.DataFrame({"Course Name": COURSES}).sem_index("Course Name", d)
        return df.sem_search("Course Name", "Cooking food", K=4, return_scores=True)  # K = N (sem_filter.py:491-497)

    a, b = both(env, run, tmp_path)
    assert "vec_scores_sim_score" in a.columns and len(a) == 4
    assert set(a["Course Name"].tolist()[:2]) == {"Cooking", "Food Sciences"}
    frames_equal(a, b)


def test_sim_join(env, tmp_path):
    def run(d):
        df1 = pd.DataFrame({"Course Name": ["History of the Atlantic World", "Riemannian Geometry"]})
        df2 = pd.DataFrame({"Skill": ["Math", "History"]}).sem_index("Skill", d)
        return df1.sem_sim_join(df2, left_on="Course Name", right_on="Skill", K=1)

    a, b = both(env, run, tmp_path)
    assert set(zip(a["Course Name"], a["Skill"])) == {("History of the Atlantic World", "History"),
                                                       ("Riemannian Geometry", "Math")}
    assert a["_scores"].dtype == np.float32
    frames_equal(a, b)


def test_sim_join_indexed_left_and_filtered_right(env, tmp_path):
    rng = np.random.default_rng(0)
    words = sum(__import__("fake_rm").TOPICS.values(), [])
    left = [" ".join(rng.choice(words, 3)) for _ in range(40)]
    right = [" ".join(rng.choice(words, 2)) for _ in range(120)]

    def run(d):
        df1 = pd.DataFrame({"L": left}).sem_index("L", d + "_l")
        df2 = pd.DataFrame({"R": right, "keep": np.arange(120) % 3 != 0}).sem_index("R", d + "_r")
        df2 = df2[df2["keep"]]  # right frame filtered AFTER indexing: ids = strict subset of the index
        return df1.sem_sim_join(df2, left_on="L", right_on="R", K=4, score_suffix="_x", keep_index=True)

    a, b = both(env, run, tmp_path)
    assert len(a) == 160 and a["keep"].all() and "_scores_x" in a.columns
    frames_equal(a, b)


def test_dedup(env, tmp_path):
    def run(d):
        df = pd.DataFrame({"Text": ["Probability and Random Processes", "Probability and Markov Chains",
                                    "Harry Potter", "Harry James Potter"]})
        return df.sem_index("Text", d).sem_dedup("Text", threshold=0.85)

    a, b = both(env, run, tmp_path)
    kept = sorted(a["Text"].tolist())
    assert len(kept) == 2 and "Harry" in kept[0] and "Probability" in kept[1]
    assert len(b) == 2 and sorted(t.split()[0] for t in b["Text"]) == ["Harry", "Probability"]


def test_filtered_vector_search_doubles_k_past_ntotal(env, tmp_path):
    def run(d):
        df = pd.DataFrame({"Course Name": ["Gourmet Cooking Advanced", "Home Cooking Basics",
                                           "Probability and Statistics", "Linear Algebra Fundamentals",
                                           "Riemannian Geometry", "History of the Atlantic World", "Harry Potter"],
                           "Category": ["Culinary", "Culinary", "Math", "Math", "Math", "History", "Fiction"]})
        df = df.sem_index("Course Name", d)
        return df[df["Category"] == "Culinary"].sem_search("Course Name", "Linear Algebra Geometry advanced", K=2)

    a, b = both(env, run, tmp_path)
    assert set(a["Category"]) == {"Culinary"} and len(a) == 2
    frames_equal(a, b)


def test_cluster_by_through_the_reference_cluster_function(env, tmp_path):
    def run(d):
        df = pd.DataFrame({"Course Name": COURSES}).sem_index("Course Name", d)
        return df.sem_cluster_by("Course Name", 2)

    a, b = both(env, run, tmp_path)
    groups = a.groupby("cluster_id")["Course Name"].apply(set).tolist()
    assert sorted(map(sorted, groups)) == [["Cooking", "Food Sciences"],
                                           ["Optimization Methods in Engineering", "Probability and Random Processes"]]
    frames_equal(a, b)


def test_float64_embeddings_and_fp16_storage(env, tmp_path):
    lotus = env["lotus"]

    def run(d):
        df = pd.DataFrame({"Course Name": COURSES}).sem_index("Course Name", d)
        return df.sem_search("Course Name", "Random Processes", K=2, return_scores=True)

    lotus.settings.configure(rm=env["rm"](np.float64), vs=env["FaissVS"]())
    a = run(str(tmp_path / "f"))
    lotus.settings.configure(rm=env["rm"](np.float64), vs=env["HipVS"]())
    b = run(str(tmp_path / "h"))
    frames_equal(a, b)
    lotus.settings.configure(rm=env["rm"](), vs=env["HipVS"](storage="fp16"))
    c = run(str(tmp_path / "h16"))
    assert c["Course Name"].tolist() == a["Course Name"].tolist()
    assert np.allclose(c["vec_scores_sim_score"], a["vec_scores_sim_score"], atol=2e-3)


# ---- cascade callers: they reach the hot path through the accessors with K = every live row ----
def _patched(env, fn, tmp_path):
    """`fn` with the reference FaissVS (unpatched accessors) and with HipVS under install(accessors=True)."""
    import lotus_amd

    lotus = env["lotus"]
    lotus.settings.configure(rm=env["rm"](), vs=env["FaissVS"]())
    a = fn(str(tmp_path / "faiss"))
    hip = env["HipVS"]()
    lotus.settings.configure(rm=env["rm"](), vs=hip)
    lotus_amd.install(accessors=True)
    try:
        b = fn(str(tmp_path / "hip"))
    finally:
        lotus_amd.uninstall()
    return a, b, hip


def _texts(n, seed):
    import fake_rm

    rng = np.random.default_rng(seed)
    words = sum(fake_rm.TOPICS.values(), [])
    return [" ".join(rng.choice(words, 3)) for _ in range(n)]


def test_sem_filter_embedding_proxy_call_uses_score_rows(env, tmp_path):
    """sem_filter.py:491-497: `df.sem_search(col, instruction, K=len(df), return_scores=True)` on a frame that was
    filtered after indexing - identical frame, and on HipVS it is served by scores() (no search call at all)."""
    texts = _texts(90, 1)

    def run(d):
        df = pd.DataFrame({"T": texts, "g": np.arange(90) % 4}).sem_index("T", d)
        df = df[df["g"] != 1]
        return df.sem_search("T", "probability of cooking history", K=len(df), return_scores=True)

    a, b, hip = _patched(env, run, tmp_path)
    assert len(a) == 67 and a["vec_scores_sim_score"].is_monotonic_decreasing
    frames_equal(a, b)
    assert not [c for c in hip.backend.calls if c[0] in ("search", "rank")]  # one score row + a host sort


def test_sem_join_cascade_helper_and_sem_topk_quick_sem(env, tmp_path):
    """sem_join.py:343-373 `run_sem_sim_join` (K = len(l2), keep_index=True, scores clipped to [0, 1]) and
    sem_topk.py:786-788 (`sem_index(...).sem_search(col, instruction, len(df))`) - identical frames."""
    l1, l2 = _texts(25, 2), _texts(60, 3)

    def run_join(d):
        import os

        from lotus.sem_ops.sem_join import run_sem_sim_join

        cwd = os.getcwd()
        os.makedirs(d, exist_ok=True)
        os.chdir(d)  # the helper indexes into a relative "<col>_index" directory
        try:
            return run_sem_sim_join(pd.Series(l1), pd.Series(l2), "left", "right")
        finally:
            os.chdir(cwd)

    a, b, _ = _patched(env, run_join, tmp_path)
    assert len(a) == 25 * 60 and a["_scores"].between(0, 1).all() and {"_left_id", "_right_id"} <= set(a.columns)
    frames_equal(a, b)

    def run_topk(d):
        df = pd.DataFrame({"T": l2})
        return df.sem_index("T", d).sem_search("T", "random markov chains", len(df))

    a, b, _ = _patched(env, run_topk, tmp_path / "t")
    assert len(a) == 60
    frames_equal(a, b)


def test_patched_accessors_keep_the_operator_cache_wrapper(env):
    import lotus_amd
    from lotus.sem_ops.sem_search import SemSearchDataframe
    from lotus.sem_ops.sem_sim_join import SemSimJoinDataframe

    before = SemSearchDataframe.__call__
    lotus_amd.install(accessors=True)
    try:
        for cls in (SemSearchDataframe, SemSimJoinDataframe):
            assert hasattr(cls.__call__, "__wrapped__")  # functools.wraps of lotus.cache.operator_cache
    finally:
        lotus_amd.uninstall()
    assert SemSearchDataframe.__call__ is before


def test_device_rm_hands_tensors_to_the_vector_store(env, tmp_path):
    """an RM whose _embed returns ONE tensor (no per-batch .cpu().numpy(), no host stack) drives
    the reference's unmodified accessors; results equal the ndarray RM's."""
    import fake_rm
    import torch

    from lotus_amd import DeviceRM

    lotus = env["lotus"]
    calls = []

    def encode(batch):
        calls.append(len(batch))
        return torch.from_numpy(fake_rm.embed(batch, np.float32))

    def run(d):
        df = pd.DataFrame({"Course Name": COURSES}).sem_index("Course Name", d)
        s = df.sem_search("Course Name", "Cooking food", K=2, return_scores=True)
        j = pd.DataFrame({"Q": ["random processes", "gourmet food"]}).sem_sim_join(df, left_on="Q", right_on="Course Name", K=2)
        return s, j

    lotus.settings.configure(rm=env["rm"](), vs=env["HipVS"]())
    s0, j0 = run(str(tmp_path / "nd"))
    rm = DeviceRM(encode, max_batch_size=3, normalize_embeddings=False)
    assert isinstance(rm, env["lotus"].models.rm.RM) if hasattr(env["lotus"], "models") else True
    lotus.settings.configure(rm=rm, vs=env["HipVS"]())
    s1, j1 = run(str(tmp_path / "dev"))
    assert calls[:2] == [3, 1]  # batched encoding into one destination tensor
    frames_equal(s0, s1)
    frames_equal(j0, j1)
    assert torch.is_tensor(rm(["a b", "c"])) and rm(["a b", "c"]).shape == (2, fake_rm.DIM)
