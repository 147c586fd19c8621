"""End-to-end on the real HIP path: HipVS + lotus_amd.ops / dedup / cluster against (i) golden frames produced by the
REFERENCE accessors (tests/golden/make_golden_frames.py) and (ii) the oracle's threshold join."""
import json
import os

import numpy as np
import pandas as pd
import pytest

import fake_rm
import oracle
import synth
from lotus_amd import RM, HipVS, _capi, ops
from lotus_amd.dedup import keep_mask, threshold_pairs

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def gold():
    with open(os.path.join(HERE, "golden", "frames_scenarios.json")) as f:
        return json.load(f)


def frame(d):
    df = pd.DataFrame(d["data"], columns=d["columns"])
    return df.set_index(d["columns"][0]).rename_axis(None)


def check(ref, got, score_cols):
    got = got.copy()
    assert ref.index.tolist() == got.index.tolist()
    assert ref.columns.tolist() == got.columns.tolist()
    for c in ref.columns:
        if c in score_cols:
            assert np.allclose(ref[c].astype(float), got[c].astype(float), atol=1e-5), c
        else:
            assert ref[c].tolist() == got[c].tolist(), c


def test_ops_match_reference_frames(hip_backend, gold, tmp_path):
    inp = gold["inputs"]
    rm = fake_rm.make_rm(RM)
    vs = HipVS(backend=hip_backend)
    df1 = pd.DataFrame({"L": inp["left"]})
    df2 = ops.sem_index(pd.DataFrame({"R": inp["right"], "keep": np.arange(400) % 5 != 2}), "R", str(tmp_path / "r"),
                        rm=rm, vs=vs)
    check(frame(gold["join_full_k3"]), ops.sem_sim_join(df1, df2, "L", "R", 3, rm=rm, vs=vs), {"_scores"})
    check(frame(gold["join_filtered_k4"]),
          ops.sem_sim_join(df1, df2[df2["keep"]], "L", "R", 4, rm=rm, vs=vs, keep_index=True, score_suffix="_s"),
          {"_scores_s"})
    check(frame(gold["search_k5"]),
          ops.sem_search(df2, "R", "optimization geometry cooking", 5, rm=rm, vs=vs, return_scores=True),
          {"vec_scores_sim_score"})
    check(frame(gold["search_filtered_k3"]),
          ops.sem_search(df2[df2["keep"]], "R", "harry potter history", 3, rm=rm, vs=vs, return_scores=True),
          {"vec_scores_sim_score"})
    dd = ops.sem_index(pd.DataFrame({"Text": inp["dedup"]}), "Text", str(tmp_path / "d"), rm=rm, vs=vs)
    assert len(ops.sem_dedup(dd, "Text", 0.9, vs=vs)) == gold["dedup_kept_count"]
    from lotus_amd.cluster import kmeans

    r = kmeans(fake_rm.embed(inp["dedup"]), 4, niter=8, backend=hip_backend)
    assert (r.assign == np.array(gold["cluster_ids"])).mean() >= 0.99


def test_range_join_matches_oracle_and_golden(hip_backend):
    z = np.load(os.path.join(HERE, "golden", "dedup_pairs.npz"))
    be = hip_backend
    packed = be.pack(z["x"], _capi.PACK_F16)
    i, j, s = threshold_pairs(be, packed, float(z["thr"]))
    up = z["pi"] < z["pj"]
    assert np.array_equal(i, z["pi"][up]) and np.array_equal(j, z["pj"][up])
    assert np.allclose(s, z["ps"][up], atol=1e-5)
    # larger random case incl. many tiles below the diagonal, both storage layouts, capacity regrowth
    x = synth.corpus(3000, 96, seed=5)
    x[1000:1400] = x[:400] + 0.05 * synth.corpus(400, 96, seed=6)
    x /= np.linalg.norm(x, axis=1, keepdims=True)
    for mode in (_capi.PACK_F16, _capi.PACK_SPLIT):
        xs = x.astype(np.float16).astype(np.float32) if mode == _capi.PACK_F16 else x
        pk = be.pack(x.astype(np.float16) if mode == _capi.PACK_F16 else x, mode)
        q, jj, ss = be.range_join(pk, pk, 0.9, q_row0=0, capacity=16)  # forces the count-then-rerun path
        got = set(zip(q.cpu().numpy().tolist(), jj.cpu().numpy().tolist()))
        sc = xs @ xs.T
        sure = {(a, b) for a, b in zip(*np.nonzero(sc > 0.9 + 2e-5)) if a < b}
        maybe = {(a, b) for a, b in zip(*np.nonzero(sc > 0.9 - 2e-5)) if a < b}
        assert sure <= got <= maybe and len(got) >= 400
    # non-self join and tile dealing: the union over phases equals the undealt result
    qx = be.pack(x[:700].astype(np.float16), _capi.PACK_F16)
    pk = be.pack(x.astype(np.float16), _capi.PACK_F16)
    full = be.range_join(pk, qx, 0.9)
    parts = [be.range_join(pk, qx, 0.9, stride=3, phase=p) for p in range(3)]
    tot = sum(len(p[0]) for p in parts)
    assert tot == len(full[0]) and tot >= 400
    u = set()
    for p in parts:
        u |= set(zip(p[0].cpu().numpy().tolist(), p[1].cpu().numpy().tolist()))
    assert u == set(zip(full[0].cpu().numpy().tolist(), full[1].cpu().numpy().tolist()))


def test_dedup_at_scale_finds_planted_duplicate_chains(hip_backend):
    """200k rows: 10k planted near-duplicates (cos ~ 0.98) incl. dup-of-dup chains + hard negatives (cos ~ 0.89)."""
    import torch

    be = hip_backend
    n0, d = 180_000, 768
    g = torch.Generator(device=be.device)
    g.manual_seed(9)
    base = torch.nn.functional.normalize(torch.randn((n0, d), generator=g, device=be.device), dim=1)

    def noisy(src, amp):
        return torch.nn.functional.normalize(
            src + amp * torch.nn.functional.normalize(torch.randn(src.shape, generator=g, device=be.device), dim=1), dim=1)

    dup1 = noisy(base[:10_000], 0.2)
    dup2 = noisy(dup1[:5_000], 0.2)      # chain: base -> dup1 -> dup2
    neg = noisy(base[10_000:15_000], 0.5)
    x = torch.cat([base, dup1, dup2, neg]).to(torch.float16)
    packed = be.pack(x, _capi.PACK_F16)
    i, j, s = threshold_pairs(be, packed, 0.95)
    assert (s > 0.95).all() and (i < j).all()
    pairs = set(zip(i.tolist(), j.tolist()))
    assert all((r, n0 + r) in pairs for r in range(0, 10_000, 97))          # base ~ dup1
    assert all((n0 + r, n0 + 10_000 + r) in pairs for r in range(0, 5_000, 97))  # dup1 ~ dup2
    assert not any(a >= n0 + 15_000 or b >= n0 + 15_000 for a, b in pairs)  # hard negatives stay out
    vals = [f"row{r}" for r in range(len(x))]
    keep = keep_mask(vals, i, j)
    assert int((~keep).sum()) == 15_000 and keep[:n0].all()  # every chain collapses onto its base row


def test_device_resident_embeddings_in_and_out(hip_backend, tmp_path):
    """Embeddings produced on the GPU are indexed and searched without a host round trip."""
    import torch

    xb = synth.corpus(5000, 128, seed=3)
    xq, planted = synth.queries(xb, 64, seed=4)
    xb_d = torch.from_numpy(xb.astype(np.float16)).to(hip_backend.device)
    xq_d = torch.from_numpy(xq.astype(np.float16)).to(hip_backend.device)
    vs = HipVS(backend=hip_backend)
    vs.index(None, xb_d, str(tmp_path / "dev"), persist=False)
    assert not os.path.exists(str(tmp_path / "dev" / "vecs"))
    out = vs(xq_d, 5, return_device=True)
    assert torch.is_tensor(out.indices) and out.indices.is_cuda
    Dr, Ir = oracle.flat_search(xb.astype(np.float16).astype(np.float32), xq.astype(np.float16).astype(np.float32), 5)
    err, hard, recall = synth.compare_topk(Dr, Ir, out.distances.cpu().numpy(), out.indices.cpu().numpy())
    assert err <= 1e-5 and hard == 0 and recall == 1.0
    host = vs(xq, 5)  # host queries against the same device-born index
    assert np.array_equal(host.indices, out.indices.cpu().numpy())
    S = vs.scores(xq_d[:4])
    assert np.allclose(S, (xq.astype(np.float16).astype(np.float32)[:4] @ xb.astype(np.float16).astype(np.float32).T),
                       atol=2e-6)
    vs2 = HipVS(backend=hip_backend)
    vs2.index(None, xb_d, str(tmp_path / "dev2"))  # persisted: loads back like any other index
    fresh = HipVS(backend=hip_backend)
    fresh.load_index(str(tmp_path / "dev2"))
    assert np.array_equal(fresh(xq, 5).indices, host.indices)


def test_device_rm_to_index_to_search_without_host_copies(hip_backend, tmp_path):
    """DeviceRM (lotus_amd/rm.py): encoder output stays a CUDA tensor through sem_index, sem_search (K = all live rows,
    served by the score-row path) and sem_sim_join; frames equal those of the ndarray RM."""
    import torch

    from lotus_amd import DeviceRM

    dev = hip_backend.device
    rm_h = fake_rm.make_rm(RM)
    rm_d = DeviceRM(lambda b: torch.from_numpy(fake_rm.embed(b)).to(dev), max_batch_size=50, normalize_embeddings=False)
    rng = np.random.default_rng(4)
    words = sum(fake_rm.TOPICS.values(), [])
    right = [" ".join(rng.choice(words, 3)) for _ in range(700)]
    left = [" ".join(rng.choice(words, 2)) for _ in range(90)]
    outs = []
    for tag, rm in (("h", rm_h), ("d", rm_d)):
        vs = HipVS(backend=hip_backend)
        df2 = ops.sem_index(pd.DataFrame({"R": right, "keep": np.arange(700) % 3 != 1}), "R", str(tmp_path / tag), rm=rm, vs=vs)
        live = df2[df2["keep"]]
        s_all = ops.sem_search(live, "R", "geometry cooking history", len(live), rm=rm, vs=vs, return_scores=True)
        j = ops.sem_sim_join(pd.DataFrame({"L": left}), live, "L", "R", 3, rm=rm, vs=vs)
        outs.append((s_all, j))
        if tag == "d":
            assert torch.is_tensor(rm(["a"])) and rm(["a"]).is_cuda
    (s0, j0), (s1, j1) = outs
    assert len(s0) == int((np.arange(700) % 3 != 1).sum()) and s0["vec_scores_sim_score"].is_monotonic_decreasing
    # ranking of ALL live rows from one score row == the search kernels' ranking (ties: lower row id first)
    vs = HipVS(backend=hip_backend)
    vs.load_index(str(tmp_path / "h"))
    ref = vs(fake_rm.embed(["geometry cooking history"]), len(s0), ids=s0.index.sort_values().tolist())
    assert np.allclose(np.asarray(ref.distances)[0], s0["vec_scores_sim_score"].to_numpy(), atol=1e-5)
    same = np.asarray(ref.indices)[0] == s0.index.to_numpy()
    assert same.mean() > 0.98  # near-ties (summation order differs between the two kernels) may swap neighbours
    check(s0, s1, {"vec_scores_sim_score"})
    check(j0, j1, {"_scores"})


def test_partial_load_from_the_row_store(hip_backend, tmp_path, monkeypatch):
    """store.py on the real backend: a rank packs only its rows out of the memory map; search results carry global ids."""
    import pickle

    xb = synth.corpus(4001, 96, seed=12).astype(np.float16)
    xq, _ = synth.queries(xb.astype(np.float32), 30, seed=2)
    d = str(tmp_path / "i")
    HipVS(backend=hip_backend).index(None, xb, d)
    monkeypatch.setattr(pickle, "load", lambda *a, **k: (_ for _ in ()).throw(AssertionError("unpickled")))
    Dr, Ir = oracle.flat_search(xb.astype(np.float32), xq.astype(np.float16).astype(np.float32), 6)
    parts = []
    for rank in range(2):
        vs = HipVS(backend=hip_backend, shard=True)
        monkeypatch.setattr(vs, "_dist", lambda r=rank: (r, 2))
        vs.load_index(d)
        ent = vs._resident[d]
        assert ent.packed.n == (2001 if rank == 0 else 2000) and ent.vecs is None
        q = hip_backend.pack(xq.astype(np.float16), _capi.PACK_F16)
        parts.append(hip_backend.search_keys(ent.packed, q, 6, 0, id_offset=ent.lo))
        assert np.array_equal(vs.get_vectors_from_index(d, [4000, 3]), xb[[4000, 3]])
    import torch

    keys = hip_backend.merge_keys(torch.stack(parts))
    D, I = hip_backend.keys_to_result(keys, 0)
    err, hard, recall = synth.compare_topk(Dr, Ir, D.cpu().numpy(), I.cpu().numpy())
    assert err <= 1e-5 and hard == 0 and recall == 1.0


@pytest.mark.parametrize("scale", [1e-2, 1.0, 1e2])
@pytest.mark.parametrize("metric", [0, 1])
def test_fp32_embeddings_of_any_magnitude_keep_fp32_accuracy(hip_backend, tmp_path, scale, metric):
    """faiss takes any float32 (faiss_vs.py:24).  HipVS stores fp32 embeddings as fp16 hi|lo pairs of x * 2^e with e chosen
    from the data, so rows of norm 1e-2 are scored as accurately as rows of norm 1e+2: error <= 1e-6 |q| |y| against a
    float64 reference (unscaled, the lo half of small values would live in fp16's subnormals)."""
    xb = (synth.corpus(30_000, 256, seed=41) * scale).astype(np.float32)
    xq = (synth.queries(xb / scale, 400, seed=42)[0] * scale).astype(np.float32)
    vs = HipVS(backend=hip_backend, metric=metric)
    vs.index(None, xb, str(tmp_path / "i"), persist=False)
    assert vs._resident[vs.index_dir].packed.exp == 6 - int(np.floor(np.log2(np.abs(xb[:262144]).max())))
    out = vs(xq, 10)
    S = xq.astype(np.float64) @ xb.astype(np.float64).T
    if metric == 1:
        S = -((xq.astype(np.float64) ** 2).sum(1)[:, None] + (xb.astype(np.float64) ** 2).sum(1)[None, :] - 2 * S)
    order = np.argsort(-S, axis=1, kind="stable")[:, :10]
    ref = np.take_along_axis(S, order, 1) * (1 if metric == 0 else -1)
    bar = 1e-6 * scale * scale * (1 if metric == 0 else 4)  # |q| |y| = scale^2; a squared distance is a sum of three such terms
    assert np.abs(out.distances - ref).max() <= bar, (np.abs(out.distances - ref).max(), bar)
    assert (out.indices == order).mean() >= 0.995
    got = vs.get_vectors_from_index(vs.index_dir, [5, 17])  # served from the device image: scale undone
    assert np.abs(got - xb[[5, 17]]).max() <= 2.0 ** -21 * np.abs(xb).max()
    sc = vs.scores(xq[:3])
    assert np.abs(sc - S[:3]).max() <= bar


def test_non_finite_and_out_of_range_inputs_are_refused_or_rescaled(hip_backend, tmp_path):
    xb = synth.corpus(5_000, 64, seed=43)
    bad = xb.copy()
    bad[77, 3] = np.inf
    vs = HipVS(backend=hip_backend)
    with pytest.raises(ValueError, match="inf or NaN"):
        vs.index(None, bad, str(tmp_path / "bad"), persist=False)
    vs.index(None, xb, str(tmp_path / "ok"), persist=False)
    q = synth.queries(xb, 8, seed=44)[0]
    qn = q.copy()
    qn[2, 5] = np.nan
    with pytest.raises(ValueError, match="inf or NaN"):
        vs(qn, 3)
    # queries a million times larger than the index's rows leave fp16's range under the index's scale: inner products are
    # searched again with an exponent of their own, same neighbours, scores scaled accordingly
    big = vs(q * 1e6, 3)
    ref = vs(q, 3)
    assert np.array_equal(big.indices, ref.indices)
    assert np.allclose(big.distances, ref.distances * 1e6, rtol=1e-5)
    # a mixed batch: only the outliers take the retry, the other queries keep the index's exponent and their accuracy
    mixed = q.copy()
    mixed[[1, 6]] *= 1e6
    factor = np.ones((8, 1), np.float32)
    factor[[1, 6]] = 1e6
    got = vs(mixed, 3)
    assert np.array_equal(got.indices, ref.indices) and np.allclose(got.distances, ref.distances * factor, rtol=1e-5)
    dev = vs(mixed, 3, return_device=True)
    assert dev.distances.is_cuda and np.array_equal(dev.indices.cpu().numpy(), ref.indices)
    assert np.allclose(dev.distances.cpu().numpy(), ref.distances * factor, rtol=1e-5)
    S = vs.scores(mixed)
    want = (q.astype(np.float64) @ xb.astype(np.float64).T) * factor
    assert np.abs(S - want).max() <= 1e-5 * factor.max()
    assert np.abs(S[[0, 2, 3, 4, 5, 7]] - want[[0, 2, 3, 4, 5, 7]]).max() <= 1e-5
    vl2 = HipVS(backend=hip_backend, metric=1)
    vl2.index(None, xb, str(tmp_path / "l2"), persist=False)
    with pytest.raises(ValueError, match="range"):
        vl2(q * 1e6, 3)
    # storage="fp16" keeps values as given: fp32 inputs beyond fp16's range are refused, not turned into inf
    v16 = HipVS(backend=hip_backend, storage="fp16")
    with pytest.raises(ValueError, match="range"):
        v16.index(None, xb * 1e6, str(tmp_path / "f16"), persist=False)


def test_kmeans_and_dedup_on_scaled_rows(hip_backend, tmp_path):
    """The pack scale is invisible to the operators: k-means centroids / objective and the threshold join come back in the
    caller's units."""
    from lotus_amd.cluster import kmeans

    rng = np.random.default_rng(45)
    c = rng.standard_normal((12, 64)).astype(np.float32)
    x = ((c[rng.integers(0, 12, 20_000)] + 0.3 * rng.standard_normal((20_000, 64))) * 1e-3).astype(np.float32)
    vs = HipVS(backend=hip_backend)
    vs.index(None, x, str(tmp_path / "km"), persist=False)
    assert vs._resident[vs.index_dir].packed.exp > 6
    r = vs.kmeans(None, 12, niter=5, return_result=True)
    hi = x.astype(np.float16)  # the oracle sees the values the device holds: hi|lo of the scaled rows, scaled back
    e = vs._resident[vs.index_dir].packed.exp
    xs = (x * np.float32(2.0 ** e)).astype(np.float32)
    stored = (xs.astype(np.float16).astype(np.float32) + (xs - xs.astype(np.float16).astype(np.float32)).astype(np.float16).astype(np.float32)) * np.float32(2.0 ** -e)
    ref = oracle.kmeans_faiss(stored, 12, niter=5)
    assert (r.assign == ref.assign).mean() >= 1 - 1e-4
    assert np.allclose(r.obj, ref.obj, rtol=1e-5) and np.allclose(r.centroids, ref.centroids, rtol=1e-4, atol=1e-9)
    xd = (synth.corpus(3_000, 64, seed=46) * 50.0).astype(np.float32)
    xd[1500:1600] = xd[:100] * (1 + 1e-3)
    vd = HipVS(backend=hip_backend)
    vd.index(None, xd, str(tmp_path / "dd"), persist=False)
    i, j, s = threshold_pairs(hip_backend, vd.packed_rows(), 0.95 * 2500.0)
    assert len(i) == 100 and np.array_equal(j - i, np.full(100, 1500)) and np.allclose(s, 2500.0 * (1 + 1e-3), rtol=1e-4)


def test_large_host_call_is_staged_and_equals_the_single_launch(hip_backend, tmp_path):
    """VS.__call__ with tens of thousands of host queries (faiss_vs.py:75 as sem_sim_join.py:132-134 calls it) goes through
    in two stages - H2D of the second stage under the first stage's search, D2H of the first stage's results under the
    second's (HipBackend.search_host_pipelined, through the bounded pinned ring of _h2d): results are per query, so the
    staged call must return exactly what one launch over all queries returns - also for queries that fail validation."""
    be = hip_backend
    xb = synth.corpus(60_000, 96, seed=41).astype(np.float16)
    xq = synth.queries(xb.astype(np.float32), 40_000, seed=42)[0].astype(np.float16)
    vs = HipVS(backend=be, storage="fp16")
    vs.index(None, xb, str(tmp_path / "idx"), persist=False)
    assert xq.shape[0] >= be.CALL_PIPELINE_MIN_QUERIES
    staged = vs(xq, 10)
    floor = be.CALL_PIPELINE_MIN_QUERIES
    try:
        be.CALL_PIPELINE_MIN_QUERIES = 1 << 40
        plain = vs(xq, 10)
    finally:
        be.CALL_PIPELINE_MIN_QUERIES = floor
    assert np.array_equal(staged.indices, plain.indices) and np.array_equal(staged.distances, plain.distances)
    Dr, Ir = oracle.flat_search(xb.astype(np.float32), xq[:2000].astype(np.float32), 10)
    err, hard, recall = synth.compare_topk(Dr, Ir, staged.distances[:2000], staged.indices[:2000], atol=1e-5)
    assert err <= 1e-5 and hard == 0 and recall >= 0.9999
    # a matrix wider than a ring slot's worth of rows per slice, float32 input rounded to fp16 storage, odd row count
    xq32 = xq[:33_001].astype(np.float32)
    a = vs(xq32, 3)
    assert np.array_equal(a.indices, plain.indices[:33_001, :3])
    bad = xq.copy()
    bad[35_000, 5] = np.inf
    with pytest.raises(ValueError):
        vs(bad, 10)
    # two threads sharing one backend: the staging ring is guarded, both get their own rows
    import threading

    outs = {}

    def run(tag, q):
        outs[tag] = vs(q, 5).indices

    t1 = threading.Thread(target=run, args=("a", xq))
    t2 = threading.Thread(target=run, args=("b", xq[::-1].copy()))
    t1.start(); t2.start(); t1.join(); t2.join()
    assert np.array_equal(outs["a"], plain.indices[:, :5]) and np.array_equal(outs["b"], plain.indices[::-1, :5])


def test_split_planner_calibrates_on_the_machine_it_runs_on(hip_backend, tmp_path):
    """`plan.calibrate(backend)`: the planner's tables measured on THIS GPU and build (seven shapes of the 100 k x 1 M join +
    the pooled-threshold shard, ~2 s), saved as the JSON `$LOTUS_AMD_PLAN_TABLES` names - `shard="auto"` is then no longer a
    constant of the box the shipped tables were fitted on."""
    import json

    from lotus_amd import plan

    path = tmp_path / "plan.json"
    try:
        t = plan.calibrate(hip_backend, save=str(path), reps=1)
        assert plan._valid(t) and 0.25 < t["base_frac"] < 0.7 and "calibrate()" in t["source"]
        assert t["loss_rows"][3] >= t["loss_rows"][1] >= 0.0  # a 125 k-row shard loses at least what a 500 k-row one does
        assert json.loads(path.read_text()) == t and plan.tables() is t
        gq, gc = plan.pick_split(8)
        assert gq * gc == 8
    finally:
        plan.use_tables(None)
