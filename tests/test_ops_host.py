"""Fast-path accessors (lotus_amd.ops) against the reference's own accessors on identical inputs, and the dedup graph
rule against the oracle restatement.  CPU only (oracle-backed test double); also writes nothing - the golden frames
used on the GPU box are produced by tests/golden/make_golden_frames.py from the same scenarios."""
import numpy as np
import pandas as pd
import pytest

import oracle
import ref_harness
import synth
from oracle_backend import OracleBackend


def test_keep_mask_equals_oracle_rule_on_random_graphs():
    from lotus_amd.dedup import component_labels, keep_mask

    rng = np.random.default_rng(0)
    for trial in range(20):
        n = int(rng.integers(5, 60))
        vals = [f"v{int(v)}" for v in rng.integers(0, max(2, n // 2), n)]  # repeated values on purpose
        m = int(rng.integers(0, 2 * n))
        i = rng.integers(0, n, m)
        j = rng.integers(0, n, m)
        ok = i < j
        i, j = i[ok], j[ok]
        # oracle rule, fed with the same pairs
        first = {}
        node = np.array([first.setdefault(v, r) for r, v in enumerate(vals)])
        a, b = node[i], node[j]
        d = a != b
        lab = oracle.dedup_components(n, a[d], b[d])
        in_pair = np.zeros(n, bool)
        in_pair[a[d]] = True
        in_pair[b[d]] = True
        expect = ~(in_pair & (lab != np.arange(n)))[node]
        assert np.array_equal(keep_mask(vals, i, j), expect)
        assert np.array_equal(component_labels(n, a[d], b[d]), lab)


def test_threshold_pairs_equals_oracle_range_join():
    from lotus_amd.dedup import threshold_pairs

    z = np.load(__import__("os").path.join(__import__("os").path.dirname(__file__), "golden", "dedup_pairs.npz"))
    be = OracleBackend()
    packed = be.pack(z["x"], 0)
    i, j, s = threshold_pairs(be, packed, float(z["thr"]))
    up = z["pi"] < z["pj"]
    assert np.array_equal(i, z["pi"][up]) and np.array_equal(j, z["pj"][up])


@pytest.mark.skipif(not ref_harness.available(), reason="reference checkout not present")
def test_fast_path_ops_equal_reference_accessors(tmp_path):
    lotus = ref_harness.import_lotus()
    from lotus.models.rm import RM

    import fake_rm
    from lotus_amd import HipVS, ops

    words = sum(fake_rm.TOPICS.values(), [])
    rng = np.random.default_rng(4)
    left = [" ".join(rng.choice(words, 3)) for _ in range(60)]
    right = [" ".join(rng.choice(words, 2)) for _ in range(200)]
    rm = fake_rm.make_rm(RM)
    vs = HipVS(backend=OracleBackend())
    lotus.settings.configure(rm=rm, vs=vs)

    df1 = pd.DataFrame({"L": left, "n": np.arange(60)})
    df2 = pd.DataFrame({"R": right, "keep": np.arange(200) % 4 != 1}).sem_index("R", str(tmp_path / "r"))
    df2f = df2[df2["keep"]]
    for other, kw in ((df2, dict(K=3)), (df2f, dict(K=5, score_suffix="_s", keep_index=True)),
                      (df2f, dict(K=2, lsuffix="_a", rsuffix="_b"))):
        ref = df1.sem_sim_join(other, left_on="L", right_on="R", **kw)
        got = ops.sem_sim_join(df1, other, "L", "R", **kw)
        pd.testing.assert_frame_equal(ref, got)

    idx1 = pd.DataFrame({"L": left}).sem_index("L", str(tmp_path / "l"))  # indexed left column
    ref = idx1.sem_sim_join(df2, left_on="L", right_on="R", K=2)
    got = ops.sem_sim_join(idx1, df2, "L", "R", 2)
    pd.testing.assert_frame_equal(ref, got)

    for frame, K in ((df2, 4), (df2f, 3), (df2f[df2f.index > 150], 500)):
        ref = frame.sem_search("R", "probability cooking", K=K, return_scores=True)
        got = ops.sem_search(frame, "R", "probability cooking", K, return_scores=True)
        pd.testing.assert_frame_equal(ref, got, check_dtype=False)

    # dedup: the reference keeps a hash-order-dependent survivor; compare what is well defined
    texts = ["Probability and Random Processes", "Probability and Markov Chains", "Harry Potter", "Harry James Potter",
             "Cooking", "Cooking", "Riemannian Geometry"]
    dd = pd.DataFrame({"Text": texts}).sem_index("Text", str(tmp_path / "d"))
    ref = dd.sem_dedup("Text", threshold=0.85)
    got = ops.sem_dedup(dd, "Text", 0.85)
    assert len(ref) == len(got) == 5
    assert sorted(t.split()[0] for t in got["Text"]) == sorted(t.split()[0] for t in ref["Text"])
    assert got["Text"].tolist() == ["Probability and Random Processes", "Harry Potter", "Cooking", "Cooking",
                                    "Riemannian Geometry"]  # first value of every group survives, equal values stay
    sub = dd[dd.index != 0]  # filtered frame: positions are remapped through the gather
    assert ops.sem_dedup(sub, "Text", 0.85)["Text"].tolist() == ["Probability and Markov Chains", "Harry Potter",
                                                                  "Cooking", "Cooking", "Riemannian Geometry"]


def test_ops_work_without_lotus_settings(tmp_path):
    """Standalone use (what runs on the GPU box, where LOTUS is absent): explicit vs= and precomputed embeddings."""
    from lotus_amd import HipVS, ops

    xb = synth.corpus(300, 32, seed=3)
    xq, planted = synth.queries(xb, 20, seed=4)
    vs = HipVS(backend=OracleBackend())
    right = ops.sem_index(pd.DataFrame({"R": [f"r{i}" for i in range(300)]}), "R", str(tmp_path / "r"), vs=vs,
                          embeddings=xb)
    left = pd.DataFrame({"L": [f"l{i}" for i in range(20)]})

    class PassThroughRM:
        def convert_query_to_query_vector(self, q):
            return xq

    out = ops.sem_sim_join(left, right, "L", "R", 2, rm=PassThroughRM(), vs=vs)
    assert out["R"].iloc[::2].tolist() == [f"r{j}" for j in planted]
    assert out["_scores"].dtype == np.float32 and len(out) == 40
    hit = ops.sem_search(right, "R", xq[:1], 3, vs=vs, return_scores=True)
    assert hit["R"].iloc[0] == f"r{planted[0]}"


@pytest.mark.skipif(not ref_harness.available(), reason="reference checkout not present")
def test_install_routes_the_accessors_and_falls_back_for_other_stores(tmp_path):
    lotus = ref_harness.import_lotus()
    from lotus.models.rm import RM
    from lotus.vector_store.faiss_vs import FaissVS

    import fake_rm
    import lotus_amd
    from lotus_amd import HipVS

    words = sum(fake_rm.TOPICS.values(), [])
    rng = np.random.default_rng(8)
    left = [" ".join(rng.choice(words, 3)) for _ in range(30)]
    right = [" ".join(rng.choice(words, 2)) + f" {i}" for i in range(90)]

    def run(d):
        df2 = pd.DataFrame({"R": right}).sem_index("R", d + "r")
        j = pd.DataFrame({"L": left}).sem_sim_join(df2, left_on="L", right_on="R", K=3)
        s = df2[df2.index % 2 == 0].sem_search("R", "history cooking", K=4, return_scores=True)
        c = df2.sem_cluster_by("R", 3, niter=5)
        dd = pd.DataFrame({"T": right + [t + " extra" for t in right[:10]]}).sem_index("T", d + "d").sem_dedup("T", 0.9)
        return j, s, c, dd

    lotus.settings.configure(rm=fake_rm.make_rm(RM), vs=FaissVS())
    ref = run(str(tmp_path / "f"))
    lotus_amd.install(accessors=True)
    try:
        from lotus.sem_ops.sem_sim_join import SemSimJoinDataframe

        assert SemSimJoinDataframe.__call__.__name__ == "sim_join"
        still_ref = run(str(tmp_path / "f2"))  # FaissVS configured: patched accessors fall through to the originals
        lotus.settings.configure(rm=fake_rm.make_rm(RM), vs=HipVS(backend=OracleBackend()))
        got = run(str(tmp_path / "h"))
    finally:
        lotus_amd.uninstall()
    assert SemSimJoinDataframe.__call__.__name__ != "sim_join"
    for a, b in ((ref, still_ref), (ref, got)):
        pd.testing.assert_frame_equal(a[0], b[0])
        pd.testing.assert_frame_equal(a[1], b[1], check_dtype=False)
        assert a[2]["cluster_id"].tolist() == b[2]["cluster_id"].tolist()
        assert len(a[3]) == len(b[3]) and 90 <= len(a[3]) <= 100


def test_ops_sem_cluster_by_standalone(tmp_path):
    from lotus_amd import HipVS, ops

    rng = np.random.default_rng(2)
    c = rng.standard_normal((3, 16)).astype(np.float32) * 5
    lab = rng.integers(0, 3, 200)
    x = (c[lab] + 0.2 * rng.standard_normal((200, 16))).astype(np.float32)
    vs = HipVS(backend=OracleBackend())
    df = ops.sem_index(pd.DataFrame({"t": [f"t{i}" for i in range(200)]}), "t", str(tmp_path / "i"), vs=vs, embeddings=x)
    out = ops.sem_cluster_by(df, "t", 3, niter=6, vs=vs)
    assert "cluster_id" in out.columns and "cluster_id" not in df.columns
    for b in range(3):
        assert len(set(out["cluster_id"][lab == b])) == 1  # every true blob maps to one cluster
    with pytest.raises(ValueError):
        ops.sem_cluster_by(df, "t", 500, vs=vs)


def test_joined_frame_shortcut_equals_the_reference_joins():
    """`ops._joined_frame` (positional takes) against the two `DataFrame.join`s of `sem_sim_join.py:152-162` on random
    frames: label indexes, overlapping column names with every suffix combination, keep_index, empty results; where
    pandas raises (overlap without suffix) the shortcut must step aside."""
    import pandas as pd

    from lotus_amd import ops

    def ref(df1, df2, left_ids, right_ids, sc, lsuffix, rsuffix, score_suffix, keep_index):
        d1, d2 = df1.copy(), df2.copy()
        d1["_left_id"] = d1.index
        d2["_right_id"] = d2.index
        temp = pd.DataFrame({"_left_id": left_ids, "_right_id": right_ids, "_scores" + score_suffix: sc})
        j = d1.join(temp.set_index("_left_id"), how="right", on="_left_id").join(
            d2.set_index("_right_id"), how="left", on="_right_id", lsuffix=lsuffix, rsuffix=rsuffix)
        if not keep_index:
            j.drop(columns=["_left_id", "_right_id"], inplace=True)
        return j

    rng = np.random.default_rng(0)
    n_equal = n_deferred = 0
    for trial in range(120):
        nl, nr = int(rng.integers(1, 8)), int(rng.integers(1, 9))
        li = rng.permutation(50)[:nl] if trial % 2 else np.arange(nl)
        ri = rng.permutation(60)[:nr] + 100 if trial % 3 else np.arange(nr)
        if trial % 7 == 0:
            li = np.array([f"L{i}" for i in li], dtype=object)
        df1 = pd.DataFrame({"q": [f"l{i}" for i in range(nl)], "x": rng.random(nl), "n": rng.integers(0, 9, nl)}, index=li)
        df2 = pd.DataFrame({"text": [f"r{i}" for i in range(nr)], "x": rng.integers(0, 9, nr).astype(np.int32)}, index=ri)
        if trial % 5 == 0:
            df2 = df2.drop(columns=["x"])
        K = int(rng.integers(1, 4))
        keep = rng.random(nl * K) > (1.0 if trial % 11 == 0 else 0.2)  # every 11th trial: an empty result
        qpos = np.repeat(np.arange(nl), K)[keep]
        left_ids = np.asarray(df1.index)[qpos]
        right_ids = np.asarray(df2.index)[rng.integers(0, nr, len(qpos))]
        sc = rng.random(len(qpos)).astype(np.float32)
        for lsuf, rsuf, ssuf, keep_index in [("", "", "", False), ("_l", "_r", "", True), ("", "_r", "_s", False), ("_l", "", "", True)]:
            got = ops._joined_frame(df1, df2, qpos, left_ids, right_ids, sc, lsuf, rsuf, ssuf, keep_index)
            try:
                want = ref(df1, df2, left_ids, right_ids, sc, lsuf, rsuf, ssuf, keep_index)
            except ValueError:
                assert got is None
                n_deferred += 1
                continue
            if got is not None:
                pd.testing.assert_frame_equal(got, want)
                n_equal += 1
    assert n_equal > 300 and n_deferred > 10
    # duplicate labels: not the shortcut's business
    dup = pd.DataFrame({"q": ["a", "b"]}, index=[1, 1])
    assert ops._joined_frame(dup, pd.DataFrame({"t": ["x"]}), np.array([0]), np.array([1]), np.array([0]),
                             np.array([0.5], np.float32), "", "", "", False) is None
