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
