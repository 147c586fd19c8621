"""How to split one sim-join over the GPUs of a node (the call being split is
``lotus/sem_ops/sem_sim_join.py:132-134`` -> ``VS.__call__``).

A join of Q queries against N corpus rows on ``world`` GPUs can be cut along either operand: ``gq`` query groups x ``gc``
corpus shards (``gq * gc == world``), every GPU searching Q / gq queries against N / gc rows; the per-shard lists are
merged inside a corpus group (one all-gather of 8-byte keys + ``lvs_merge_keys``) and concatenated across the query
groups (one all-gather, no merge).  ``gc == world`` is BASELINE's row split, ``gq == world`` the query split.

The total MFMA work is the same for every split; what differs is how far the fused top-k kernel runs below its
long-stream rate on the per-GPU shape.  Measured on one MI355X (bench.py legs ``node_plan_8gpu`` / ``shard_*``;
fraction of the dense fp16 MFMA roof, profiles/r03a_bench.json, a box whose 100 k x 1 M launch runs at 44.0 %):

    per-GPU shape at 8 GPUs     1 x 8: 100 k x 125 k  35.8 %      2 x 4: 50 k x 250 k  37.0 %
                                4 x 2: 25 k x 500 k   36.8 %      8 x 1: 12.5 k x 1 M  36.8 %
    row split at 4 / 2 GPUs     100 k x 250 k  39.6 %             100 k x 500 k  42.3 %

Halving the corpus stream costs threshold events per flop (the top-k slow path: ~ln(N) / N), halving the query count
costs L2 sharing of a corpus stream and fuller tail rounds; the two penalties are about equal per halving and mildly
convex, so the balanced splits come out ~3 % ahead of the pure row split (4.81 M vs 4.66 M q/s kernel-side at 8 GPUs) - a
small but free gain, as long as the larger corpus shard still fits.  The corpus side has to fit: a shard must leave
room in HBM for the queries and workspaces.
"""
from __future__ import annotations

import math

# fraction of the MFMA roof lost after h halvings of the per-GPU corpus stream / query count relative to 100 k x 1 M
# (measured, see above; linear interpolation between the points, extrapolated with the last slope)
_LOSS_PER_HALVING = (0.0, 0.017, 0.044, 0.078, 0.12)
_BASE_FRAC = 0.44
_REF_QUERIES, _REF_ROWS = 100_000, 1_000_000
HBM_BYTES = 288e9


def _loss(halvings: float) -> float:
    h = max(0.0, halvings)
    i = int(h)
    t = _LOSS_PER_HALVING
    if i + 1 < len(t):
        return t[i] + (h - i) * (t[i + 1] - t[i])
    return t[-1] + (h - (len(t) - 1)) * (t[-1] - t[-2])


def projected_fraction(queries_per_gpu: float, rows_per_gpu: float) -> float:
    """Projected fraction of the MFMA roof of the fused top-k kernel on one GPU's share of a join."""
    hq = math.log2(_REF_QUERIES / max(1.0, queries_per_gpu))
    hn = math.log2(_REF_ROWS / max(1.0, rows_per_gpu))
    return max(0.05, _BASE_FRAC - _loss(hq) - _loss(hn))


def splits(world: int):
    return [(gq, world // gq) for gq in range(1, world + 1) if world % gq == 0]


def pick_split(world: int, nq: int = _REF_QUERIES, nb: int = _REF_ROWS, d: int = 768, bytes_per_value: int = 2,
               hbm_bytes: float = HBM_BYTES):
    """-> (gq, gc) with the best projected node throughput among the splits whose corpus shard fits one GPU's HBM
    (a shard may take at most 60 % of it).  Ties go to the split with more corpus shards (less HBM per GPU)."""
    best, best_f = (1, world), -1.0
    for gq, gc in splits(world):
        if (nb / gc) * d * bytes_per_value > 0.6 * hbm_bytes:
            continue
        f = projected_fraction(nq / gq, nb / gc)
        if f > best_f + 1e-9 or (abs(f - best_f) <= 1e-9 and gc > best[1]):
            best, best_f = (gq, gc), f
    return best
