"""Host-side pieces added in round 3: the regenerable bench inputs (SURVEY.md 8(d) recipe) and the split planner."""
import numpy as np

import benchdata
from lotus_amd import plan


def test_inputs_are_a_function_of_config_and_block_only(monkeypatch):
    monkeypatch.setattr(benchdata, "BLOCK_ROWS", 3000)
    a = benchdata.corpus(3, 7001, 24, threads=1)
    monkeypatch.setattr(benchdata, "CHUNK_ROWS", 333)      # how a block's draws are split does not change the values
    b = benchdata.corpus(3, 7001, 24, threads=4)
    assert a.dtype == np.float16 and np.array_equal(a, b)
    assert np.array_equal(benchdata.corpus(3, 7001, 24, rows=(2500, 6100)), a[2500:6100])  # a rank's slice: whole blocks drawn
    assert not np.array_equal(benchdata.corpus(4, 7001, 24)[:100], a[:100])                # another config: another stream
    assert np.abs(np.linalg.norm(a.astype(np.float32), axis=1) - 1).max() < 2e-3
    rng = np.random.default_rng(np.random.SeedSequence([benchdata.SEED, 3, 1]))            # block 1 = rows 3000..5999
    x = rng.standard_normal((3000, 24), dtype=np.float32)
    x /= np.sqrt(np.einsum("ij,ij->i", x, x))[:, None]
    assert np.array_equal(a[3000:6000], x.astype(np.float16))
    q, j = benchdata.queries(3, a, 50)
    cos = (q.astype(np.float32) * a[j].astype(np.float32)).sum(1)
    assert cos.min() > 0.5 and np.array_equal(benchdata.queries(3, a, 50)[0], q)


def test_planted_duplicates_are_exactly_the_pairs_above_the_threshold():
    x, plants = benchdata.dedup_rows(4, 6000, 96)
    n_base, n_dup, n_chain, n_neg = benchdata.dedup_layout(6000)
    assert n_base + n_dup + n_chain + n_neg == 6000 and len(plants["row"]) == n_dup + n_chain + n_neg
    assert (plants["src"] < plants["row"]).all()
    sure, maybe = benchdata.dedup_expected_pairs(x, plants, 0.95)
    x32 = x.astype(np.float32)
    S = x32 @ x32.T
    iu = np.triu_indices(6000, 1)
    got = set(zip(iu[0][S[iu] > 0.95 + 2e-5].tolist(), iu[1][S[iu] > 0.95 + 2e-5].tolist()))
    assert got <= (sure | maybe) and sure <= set(zip(iu[0][S[iu] > 0.95 - 2e-5].tolist(), iu[1][S[iu] > 0.95 - 2e-5].tolist()))
    assert len(sure) >= n_dup + n_chain  # every direct duplicate and every chain link


def test_blobs_carry_their_labels():
    x, lab = benchdata.blobs(5, 5000, 64, 20)
    c = benchdata.blob_centres(5, 20, 64)
    assert ((x.astype(np.float32) @ c.T).argmax(1) == lab).mean() > 0.999


def test_split_planner():
    plan.use_tables(dict(plan._BUILTIN))  # the model's behaviour on a FIXED set of numbers (round 4's fit)
    try:
        assert plan.splits(8) == [(1, 8), (2, 4), (4, 2), (8, 1)]
        for w in (1, 2, 4, 8):
            gq, gc = plan.pick_split(w)
            assert gq * gc == w
        assert plan.pick_split(8) == (8, 1)                       # configs[2] fits every GPU: the query split runs ~2 points faster
        assert plan.pick_split(2) in ((1, 2), (2, 1))
        assert plan.pick_split(8, nq=100_000, nb=400_000_000) == (1, 8)   # 614 GB of corpus: only the row split fits
        assert plan.pick_split(8, nq=1_000_000, nb=8_000)[0] > 1  # a corpus of a few tiles: split the queries instead
        f = plan.projected_fraction
        assert f(100_000, 1_000_000) > f(100_000, 125_000) > f(1_000, 125_000)
    finally:
        plan.use_tables(None)


def test_planner_tables_are_data(tmp_path, monkeypatch):
    """The planner's numbers are data with a provenance: the shipped JSON (re-fitted from a bench.py line by
    tools/refit_plan.py), a file named by $LOTUS_AMD_PLAN_TABLES (what `plan.calibrate(backend, save=...)` writes on the machine
    at hand), the built-in fit as the last resort; a bench.py line's legs turn into tables by the documented arithmetic."""
    import json

    shipped = plan.load_tables()
    assert plan._valid(shipped) and "source" in shipped and shipped["loss_rows"][0] == 0.0
    for w in (2, 4, 8):  # whatever the shipped numbers are: a legal split, and BASELINE's corpus still fits the row split
        gq, gc = plan.pick_split(w)
        assert gq * gc == w
        assert plan.pick_split(w, nb=400_000_000 * w // 8 if w == 8 else 1_000_000)[0] * gc >= 1
    line = {"roofline": {"frac": 0.44, "csrc_sha": "abc"},
            "legs": {"shard_100k_x_500k": {"frac": 0.425}, "shard_100k_x_250k": {"frac": 0.405},
                     "node_plan_8gpu": {"splits": {"1x8": {"frac": 0.38}, "2x4": {"frac": 0.40}, "4x2": {"frac": 0.41},
                                                   "8x1": {"frac": 0.43}}},
                     "world8_rehearsal": {"frac": 0.42, "kernel_ms_per_shard": 18.0, "seed_pass_ms_per_shard": 0.72}}}
    t = plan.tables_from_bench(line)
    assert t["base_frac"] == 0.44 and t["loss_rows"][:4] == [0.0, 0.015, 0.035, 0.06] and t["loss_queries"][3] == 0.01
    assert abs(t["pooled_gain_per_halving"] - (0.42 * 18.0 / 18.72 - 0.38) / 3) < 1e-4 and "abc" in t["source"]
    path = tmp_path / "mine.json"
    path.write_text(json.dumps(dict(t, base_frac=0.30)))
    monkeypatch.setenv("LOTUS_AMD_PLAN_TABLES", str(path))
    plan.use_tables(None)
    try:
        assert plan.tables()["base_frac"] == 0.30 and abs(plan.projected_fraction(100_000, 1_000_000) - 0.30) < 1e-9
        path.write_text("{not json")
        plan.use_tables(None)
        assert plan.tables()["source"] == shipped["source"]  # an unreadable file falls through to the shipped tables
    finally:
        monkeypatch.delenv("LOTUS_AMD_PLAN_TABLES")
        plan.use_tables(None)


def test_bench_cpu_baseline_reports_a_parity_sample_of_the_timed_run():
    """bench.py's CPU leg times the comparators AND holds the GPU result against their output (here the oracle stands in for
    the GPU result): the helper must keep working without a GPU, it runs on the driver's box after the timed region."""
    import bench
    import oracle
    import synth

    xb = synth.corpus(70_000, 48, seed=4).astype(np.float16)
    xq = synth.queries(xb.astype(np.float32), 300, seed=5)[0].astype(np.float16)
    Dg, Ig = oracle.flat_search(xb.astype(np.float32), xq.astype(np.float32), 10, 0)
    out = bench.cpu_baseline(np, xb, xq, 300, 10, Dg, Ig)
    assert out["kind"] == "port" and out["value"] > 0 and out["cores"] >= 1
    par = out["gpu_parity_on_the_timed_sample"]
    assert par["recall_at_k"] == 1.0 and par["id_mismatches_outside_near_ties"] == 0 and par["max_abs_score_err"] <= 1e-5
    wrong = Ig.copy()
    wrong[7, 0] = (wrong[7, 0] + 12345) % len(xb)  # a planted neighbour replaced by an unrelated row: must be counted
    bad = bench.cpu_baseline(np, xb, xq, 300, 10, Dg, wrong)["gpu_parity_on_the_timed_sample"]
    assert bad["id_mismatches_outside_near_ties"] >= 1 and bad["recall_at_k"] < 1.0


def test_bench_line_keeps_the_contract_keys_and_fits_the_record():
    """bench.py prints a compacted line (5 significant digits, leg configuration keys and notes to stderr): the contract's keys
    must survive the compaction, the legs' figures too, and the line must stay below 6 KB so that the driver's 8 KB tail of
    stdout holds all of it.  Replayed on the verbose form of a real run (profiles/r05x_bench_details.txt)."""
    import json
    import os
    import re

    import bench

    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "r05x_bench_details.txt")
    full = json.loads(re.search(r"BENCH_DETAILS (\{.*\})", open(path).read()).group(1))["line"]
    line = json.dumps(bench.sig(bench.compact(full)))
    assert len(line) < 6000, len(line)
    out = json.loads(line)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline", "cpu_baseline", "legs", "legs_summary"):
        assert key in out, key
    assert "workload" in out["config"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in out["roofline"], key
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert key in out["cpu_baseline"], key
    assert isinstance(out["cpu_baseline"]["cores"], int)
    assert list(out)[-1] == "legs_summary"  # printed last: a truncated record still ends with the figures that matter
    legs = out["legs"]
    assert {"cfg2_10k_x_1M", "q1", "q128", "q1024", "node_plan_8gpu", "world8_rehearsal", "t_call_host_to_host",
            "range_selfjoin_cfg4", "kmeans_parity_mode", "kmeans_full_iter_10M_x_1024", "fp32_join_100k_x_1M"} <= set(legs)
    assert "frac" in legs["q128"] and "kernel_ms" in legs["q128"] and "note" not in legs["world8_rehearsal"]
    assert legs["kmeans_parity_mode"]["all_flips_are_near_ties"] in (True, False) and "seconds" in legs["kmeans_parity_mode"]
    assert bench.sig(0.123456789) == 0.12346 and bench.sig({"a": [1.0000001, 2]}) == {"a": [1.0, 2]}


def test_bench_builds_the_torchrun_command_for_its_own_ranks(monkeypatch):
    """`python bench.py --gpus 4` without a launcher re-executes itself under torch.distributed.run (one rank per GPU,
    127.0.0.1, a free port) with the same arguments; under a launcher (WORLD_SIZE set) it does not."""
    import subprocess
    import sys
    import types

    import bench

    seen = {}

    def fake_run(cmd, env=None, **kw):
        seen["cmd"], seen["env"] = cmd, env
        return types.SimpleNamespace(returncode=0)

    monkeypatch.setattr(subprocess, "run", fake_run)
    monkeypatch.setenv("LOTUS_BENCH_REHEARSAL", "1")  # no GPU count check on this CPU-only box
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3", "--warmup", "1"])
    rc = bench.self_launch(bench.parse())
    assert rc == 0
    cmd = seen["cmd"]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nproc-per-node=4" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert 1024 < int(cmd[cmd.index("--master-port") + 1]) < 65536
    assert cmd[-6:] == ["--gpus", "4", "--steps", "3", "--warmup", "1"] and cmd[-7].endswith("bench.py")
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0" and int(seen["env"]["OMP_NUM_THREADS"]) >= 1
