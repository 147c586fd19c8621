"""How to split one sim-join over the GPUs of a node (the call being split is
``lotus/sem_ops/sem_sim_join.py:132-134`` -> ``VS.__call__``).

A join of Q queries against N corpus rows on ``world`` GPUs can be cut along either operand: ``gq`` query groups x ``gc``
corpus shards (``gq * gc == world``), every GPU searching Q / gq queries against N / gc rows; the per-shard lists are
merged inside a corpus group (one all-gather of 8-byte keys + ``lvs_merge_keys``) and concatenated across the query
groups (one all-gather, no merge).  ``gc == world`` is BASELINE's row split, ``gq == world`` the query split.

The total MFMA work is the same for every split; what differs is how far the fused top-k kernel runs below its
long-stream rate on the per-GPU shape.  Measured on one MI355X (bench.py legs ``node_plan_8gpu`` / ``shard_*`` /
``world8_rehearsal``, profiles/r05a_bench.json and profiles/r05d_plan_sweep.log; fraction of the dense fp16 MFMA roof, the
100 k x 1 M launch at 45.1 % on that box; round 4: remainder query tiles in groups of their own shape, pooled sample
thresholds for corpus shards, wide groups for ~100 query tiles on short slabs):

    per-GPU shape at 8 GPUs     1 x 8: 100 k x 125 k  38.8 % own thresholds, 41.2 % with all shards' samples pooled
                                2 x 4: 50 k x 250 k   41.3 %     4 x 2: 25 k x 500 k  41.3 %     8 x 1: 12.5 k x 1 M  43.3 %
    row split at 4 / 2 GPUs     100 k x 250 k  41.4 %            100 k x 500 k  43.6 %
    fewer queries, whole corpus 50 k / 25 k / 12.5 k x 1 M  43.5 / 43.1 / 43.3 %

Halving the corpus stream costs threshold events per flop (the top-k slow path: ~ln(N) / N, mildly convex: 1.5, 3.7, 6.3
points after 1, 2, 3 halvings with a shard's own thresholds; pooling the sample thresholds of all corpus shards gives ~0.8
points back per halving); halving the query count costs next to nothing since the remainder query tiles stopped idling
CUs (1.5-2 points, flat).  So with the corpus small enough to replicate, the query split (8 x 1) is ahead by ~2 points
and ``"auto"`` takes it; the row split (BASELINE's configuration, ``shard=True``) is the one that scales the corpus past
one GPU's HBM and stays the explicit default of ``HipVS(shard=True)``.  Projections closer than a point are a tie, which goes
to the split with the most corpus shards (least HBM per GPU).  The planner ranks by the projected fraction, so a per-GPU
shape that falls off a cliff (a few hundred queries per GPU, a corpus shard of a few tiles) loses.
"""
from __future__ import annotations

import json
import math
import os

# ---- the model's numbers are DATA, not code (round 5) --------------------------------------------------------------------
# points of the MFMA roof lost after h halvings of the per-GPU corpus stream / query count relative to 100 k x 1 M (linear
# interpolation between the points, extrapolated with the last slope), the gain per halving from pooled sample thresholds,
# the long-stream fraction itself.  They come, in this order, from
#   1. the file named by $LOTUS_AMD_PLAN_TABLES (e.g. one written by `calibrate()` on the machine at hand),
#   2. lotus_amd/plan_tables.json - shipped, re-fitted by tools/refit_plan.py from a bench.py line (its `source` says which
#      run, build and device),
#   3. the built-in values below (round 4's fit, profiles/r05a_bench.json + r05d_plan_sweep.log).
# `calibrate(backend)` measures the seven shapes the tables need on the GPU it is given (~2 s) and returns / saves tables
# for THIS machine and build; `tables_from_bench(line)` does the same arithmetic on a bench.py line's legs.
_BUILTIN = {
    "loss_rows": [0.0, 0.015, 0.037, 0.063, 0.10],
    "loss_queries": [0.0, 0.015, 0.020, 0.018, 0.03],
    "pooled_gain_per_halving": 0.008,
    "base_frac": 0.451,
    "source": "built-in: round 4 fit (profiles/r05a_bench.json, profiles/r05d_plan_sweep.log), d = 768, k = 10",
}
_REF_QUERIES, _REF_ROWS = 100_000, 1_000_000
_TIE = 0.005  # projected fractions closer than this are a tie (the legs of one run agree to ~0.3 points)
HBM_BYTES = 288e9
TABLES_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "plan_tables.json")


def _valid(t: dict) -> bool:
    try:
        return (len(t["loss_rows"]) >= 2 and len(t["loss_queries"]) >= 2 and 0.05 < float(t["base_frac"]) <= 1.0
                and all(-0.05 <= float(v) <= 0.9 for v in list(t["loss_rows"]) + list(t["loss_queries"]))
                and -0.05 <= float(t["pooled_gain_per_halving"]) <= 0.1)
    except (KeyError, TypeError, ValueError):
        return False


def load_tables(path: str | None = None) -> dict:
    """The tables `pick_split` uses: `path`, else $LOTUS_AMD_PLAN_TABLES, else the shipped JSON, else the built-in fit."""
    for cand in (path, os.environ.get("LOTUS_AMD_PLAN_TABLES"), TABLES_PATH):
        if not cand:
            continue
        try:
            with open(cand) as fp:
                t = json.load(fp)
        except (OSError, ValueError):
            continue
        if _valid(t):
            return t
    return dict(_BUILTIN)


_tables = None


def tables() -> dict:
    global _tables
    if _tables is None:
        _tables = load_tables()
    return _tables


def use_tables(t: dict | None) -> None:
    """Install tables (e.g. the result of `calibrate`); None re-reads the files."""
    global _tables
    if t is not None and not _valid(t):
        raise ValueError("not a plan table")
    _tables = t


def tables_from_fractions(base: float, rows: dict, queries3: float, mixed: dict, pooled3: float | None, source: str) -> dict:
    """Tables from measured fractions of the MFMA roof: `base` 100 k x 1 M; `rows[h]` 100 k x (1 M / 2^h) with a shard's own
    thresholds, h = 1..3; `queries3` 12.5 k x 1 M; `mixed` {(1, 2): 50 k x 250 k, (2, 1): 25 k x 500 k} (query halvings, row
    halvings) - the one- and two-halving query losses follow from them by additivity; `pooled3` the 125 k-row shard with
    pooled thresholds."""
    lr = [0.0] + [max(0.0, base - rows[h]) for h in (1, 2, 3)]
    lr.append(lr[3] + max(lr[3] - lr[2], 0.0) * 1.4)
    lq3 = max(0.0, base - queries3)
    lq1 = min(max(0.0, base - mixed[(1, 2)] - lr[2]), max(lq3, 0.03)) if (1, 2) in mixed else lq3 / 2
    lq2 = min(max(0.0, base - mixed[(2, 1)] - lr[1]), max(lq3, 0.03)) if (2, 1) in mixed else lq3 * 0.8
    lq = [0.0, lq1, lq2, lq3, lq3 + max(lq3 - lq2, 0.004) * 1.5]
    gain = 0.0 if pooled3 is None else max(0.0, (pooled3 - rows[3]) / 3.0)
    return {"loss_rows": [round(v, 4) for v in lr], "loss_queries": [round(v, 4) for v in lq],
            "pooled_gain_per_halving": round(gain, 4), "base_frac": round(base, 4), "source": source}


def tables_from_bench(line: dict, source: str = "") -> dict:
    """The same from a bench.py line (legs `node_plan_8gpu`, `shard_100k_x_500k`, `shard_100k_x_250k`, `world8_rehearsal`)."""
    legs = line["legs"]
    sp = legs["node_plan_8gpu"]["splits"]
    rows = {1: legs["shard_100k_x_500k"]["frac"], 2: legs["shard_100k_x_250k"]["frac"], 3: sp["1x8"]["frac"]}
    mixed = {(1, 2): sp["2x4"]["frac"], (2, 1): sp["4x2"]["frac"]}
    w8 = legs.get("world8_rehearsal") or {}
    pooled = None
    if w8.get("frac") and w8.get("kernel_ms_per_shard"):  # the sample pass is part of the price of pooled thresholds
        pooled = w8["frac"] * w8["kernel_ms_per_shard"] / (w8["kernel_ms_per_shard"] + w8.get("seed_pass_ms_per_shard", 0.0))
    src = source or f"bench.py line, csrc {line.get('roofline', {}).get('csrc_sha', '?')}"
    return tables_from_fractions(line["roofline"]["frac"], rows, sp["8x1"]["frac"], mixed, pooled, src)


def calibrate(backend, d: int = 768, k: int = 10, save: str | None = None, reps: int = 2) -> dict:
    """Measure the tables on THIS machine: the fused top-k kernel on device-generated unit rows at the reference shape and at
    1-3 halvings of either operand (seven shapes + the pooled-threshold shard, about two seconds of GPU time), timed with the
    library's own events.  -> tables (also installed for this process; `save`: write them as JSON for $LOTUS_AMD_PLAN_TABLES)."""
    import torch

    from . import _capi

    be = backend
    g = torch.Generator(device=be.device)
    g.manual_seed(1234)

    def unit(n):
        out = torch.empty((n, d), dtype=torch.float16, device=be.device)
        for r0 in range(0, n, 1 << 18):
            r1 = min(n, r0 + (1 << 18))
            out[r0:r1] = torch.nn.functional.normalize(torch.randn((r1 - r0, d), generator=g, device=be.device), dim=1).half()
        return be.pack(out, _capi.PACK_F16)

    corpus, queries = unit(_REF_ROWS), unit(_REF_QUERIES)

    def frac(nq, nb, seeds=None, extra_ms=0.0):
        cq, cb = be.slice_rows(queries, 0, nq), be.slice_rows(corpus, 0, nb)
        be.search_keys(cb, cq, k, _capi.METRIC_IP, seed_scores=seeds)
        be.synchronize()
        be.timing_enable(True)
        for _ in range(reps):
            be.search_keys(cb, cq, k, _capi.METRIC_IP, seed_scores=seeds)
        be.synchronize()
        tot, cnt = be.timing_read()
        be.timing_enable(False)
        return 2.0 * nq * nb * d / ((tot / max(cnt, 1) + extra_ms) * 1e-3) / 2.5e15

    Q, N = _REF_QUERIES, _REF_ROWS
    base = frac(Q, N)
    rows = {h: frac(Q, N >> h) for h in (1, 2, 3)}
    mixed = {(1, 2): frac(Q >> 1, N >> 2), (2, 1): frac(Q >> 2, N >> 1)}
    q3 = frac(Q >> 3, N)
    per = N >> 3
    tiles = be.seed_tiles(Q, per, k, _capi.PACK_F16, _capi.PACK_F16)
    pooled = None
    if tiles:
        sample = lambda: torch.cat([be.seed_scores(be.slice_rows(corpus, r * per, (r + 1) * per), queries, _capi.METRIC_IP, tiles)
                                    for r in range(8)])
        seeds = sample()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        sample()
        e1.record()
        e1.synchronize()
        pooled = frac(Q, per, seeds, extra_ms=e0.elapsed_time(e1) / 8)  # the sample pass is part of the price
    name = torch.cuda.get_device_name(be.device) if be.device.type == "cuda" else str(be.device)
    t = tables_from_fractions(base, rows, q3, mixed, pooled, f"calibrate() on {name}, d = {d}, k = {k}")
    use_tables(t)
    if save:
        with open(save, "w") as fp:
            json.dump(t, fp, indent=1)
    return t


def _loss(table, halvings: float) -> float:
    h = max(0.0, halvings)
    i = int(h)
    if i + 1 < len(table):
        return table[i] + (h - i) * (table[i + 1] - table[i])
    return table[-1] + (h - (len(table) - 1)) * (table[-1] - table[-2])


def projected_fraction(queries_per_gpu: float, rows_per_gpu: float) -> float:
    """Projected fraction of the MFMA roof of the fused top-k kernel on one GPU's share of a join."""
    t = tables()
    hq = math.log2(_REF_QUERIES / max(1.0, queries_per_gpu))
    hn = math.log2(_REF_ROWS / max(1.0, rows_per_gpu))
    pooled = float(t["pooled_gain_per_halving"]) * max(0.0, min(hn, 3.0)) if queries_per_gpu >= 2048 else 0.0
    return max(0.05, float(t["base_frac"]) - _loss(t["loss_queries"], hq) - _loss(t["loss_rows"], hn) + pooled)


def splits(world: int):
    return [(gq, world // gq) for gq in range(1, world + 1) if world % gq == 0]


def pick_split(world: int, nq: int = _REF_QUERIES, nb: int = _REF_ROWS, d: int = 768, bytes_per_value: int = 2,
               hbm_bytes: float = HBM_BYTES):
    """-> (gq, gc): the split with the best projected node throughput among those whose corpus shard fits one GPU's HBM
    (a shard may take at most 60 % of it); projections within a point of the best are a tie, which goes to the split
    with the most corpus shards (least HBM per GPU)."""
    cands = []
    for gq, gc in splits(world):
        if (nb / gc) * d * bytes_per_value > 0.6 * hbm_bytes:
            continue
        cands.append((projected_fraction(nq / gq, nb / gc), gq, gc))
    if not cands:
        return (1, world)
    best = max(f for f, _, _ in cands)
    _, gq, gc = max((c for c in cands if c[0] >= best - _TIE), key=lambda c: c[2])
    return (gq, gc)
