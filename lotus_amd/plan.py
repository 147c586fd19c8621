"""How to split one sim-join over the GPUs of a node (the call being split is
``lotus/sem_ops/sem_sim_join.py:132-134`` -> ``VS.__call__``).

A join of Q queries against N corpus rows on ``world`` GPUs can be cut along either operand: ``gq`` query groups x ``gc``
corpus shards (``gq * gc == world``), every GPU searching Q / gq queries against N / gc rows; the per-shard lists are
merged inside a corpus group (one all-gather of 8-byte keys + ``lvs_merge_keys``) and concatenated across the query
groups (one all-gather, no merge).  ``gc == world`` is BASELINE's row split, ``gq == world`` the query split.

The total MFMA work is the same for every split; what differs is how far the fused top-k kernel runs below its
long-stream rate on the per-GPU shape.  Measured on one MI355X (bench.py legs ``node_plan_8gpu`` / ``shard_*``; fraction
of the dense fp16 MFMA roof, profiles/r04c_bench.json - thresholds seeded from a sample, 100 k x 1 M at 45.8 %; in
brackets the two boxes of profiles/r03a_bench.json / r03j_bench.json with cold lists, 44.0 / 44.6 % there):

    per-GPU shape at 8 GPUs     1 x 8: 100 k x 125 k  39.1 % (35.8 / 37.5)     2 x 4: 50 k x 250 k  38.8 % (37.0 / 37.4)
                                4 x 2: 25 k x 500 k   38.9 % (36.8 / 37.1)     8 x 1: 12.5 k x 1 M  39.1 % (36.8 / 36.4)
    row split at 4 / 2 GPUs     100 k x 250 k  41.8 % (39.6 / 40.1)            100 k x 500 k  43.9 % (42.3 / 43.0)

Halving the corpus stream costs threshold events per flop (the top-k slow path: ~ln(N) / N, mildly convex: 1.9, 4.0, 6.7
points after 1, 2, 3 halvings); halving the query count costs L2 sharing of a corpus stream and fuller tail rounds (3.0,
5.0, 6.7 points, mildly concave).  The sums come out within a point of each other - inside the box-to-box spread - so the
kernel gives no reason to prefer a split, and the tie goes to the one with the most corpus shards: least HBM per GPU, and
the only split that scales the corpus past one GPU (BASELINE's configuration).  The planner still ranks by the projected
fraction, so a per-GPU shape that falls off a cliff (a few hundred queries per GPU, a corpus shard of a few tiles) loses.
"""
from __future__ import annotations

import math

# points of the MFMA roof lost after h halvings of the per-GPU corpus stream / query count relative to 100 k x 1 M
# (measured, see above; linear interpolation between the points, extrapolated with the last slope)
_LOSS_ROWS = (0.0, 0.019, 0.040, 0.067, 0.10)
_LOSS_QUERIES = (0.0, 0.030, 0.050, 0.067, 0.09)
_BASE_FRAC = 0.458
_REF_QUERIES, _REF_ROWS = 100_000, 1_000_000
_TIE = 0.01  # projected fractions closer than this are a tie (box-to-box spread of the measurements)
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
    return max(0.05, _BASE_FRAC - _loss(_LOSS_QUERIES, hq) - _loss(_LOSS_ROWS, hn))


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
