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

import math

# points of the MFMA roof lost after h halvings of the per-GPU corpus stream / query count relative to 100 k x 1 M
# (measured, see above; linear interpolation between the points, extrapolated with the last slope)
_LOSS_ROWS = (0.0, 0.015, 0.037, 0.063, 0.10)
_POOLED_GAIN_PER_HALVING = 0.008   # corpus shards start from the pooled sample thresholds of all shards (HipVS._pooled_seed_scores)
_LOSS_QUERIES = (0.0, 0.015, 0.020, 0.018, 0.03)
_BASE_FRAC = 0.451
_REF_QUERIES, _REF_ROWS = 100_000, 1_000_000
_TIE = 0.005  # projected fractions closer than this are a tie (the legs of one run agree to ~0.3 points)
HBM_BYTES = 288e9


def _loss(table, halvings: float) -> float:
    h = max(0.0, halvings)
    i = int(h)
    if i + 1 < len(table):
        return table[i] + (h - i) * (table[i + 1] - table[i])
    return table[-1] + (h - (len(table) - 1)) * (table[-1] - table[-2])


def projected_fraction(queries_per_gpu: float, rows_per_gpu: float) -> float:
    """Projected fraction of the MFMA roof of the fused top-k kernel on one GPU's share of a join."""
    hq = math.log2(_REF_QUERIES / max(1.0, queries_per_gpu))
    hn = math.log2(_REF_ROWS / max(1.0, rows_per_gpu))
    pooled = _POOLED_GAIN_PER_HALVING * max(0.0, min(hn, 3.0)) if queries_per_gpu >= 2048 else 0.0
    return max(0.05, _BASE_FRAC - _loss(_LOSS_QUERIES, hq) - _loss(_LOSS_ROWS, hn) + pooled)


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
