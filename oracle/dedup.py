"""Oracle: ``sem_dedup`` arithmetic.  Test infrastructure only.

Follows ``lotus/sem_ops/sem_dedup.py:45-91``: self sim-join with K = N, keep pairs whose score is strictly
greater than the threshold (:46) and whose values differ (:47,54), connected components over the VALUES
(:58-84), keep the first member of each component and drop the rest (:87-91).

The reference picks each component's survivor from the iteration order of a Python ``set`` of string tuples
(hash-seed dependent); the oracle fixes it to the lowest row position, and parity is defined on the component
partition and the kept-row count (SURVEY.md section 7 "Hard parts").
"""
from __future__ import annotations

import numpy as np

from .flat import as_f32


def range_self_join(x, threshold: float, block: int = 2048):
    """All ordered pairs (i, j), i != j, with <x_i, x_j> > threshold (float32 sgemm scores).

    Returns (i int64, j int64, score float32) sorted by (i, j)."""
    x = as_f32(x)
    n = x.shape[0]
    thr = np.float32(threshold)
    ii, jj, ss = [], [], []
    for a0 in range(0, n, block):
        a1 = min(n, a0 + block)
        s = x[a0:a1] @ x.T
        r, c = np.nonzero(s > thr)
        keep = (r + a0) != c
        ii.append(r[keep] + a0)
        jj.append(c[keep])
        ss.append(s[r[keep], c[keep]])
    i = np.concatenate(ii) if ii else np.zeros(0, np.int64)
    j = np.concatenate(jj) if jj else np.zeros(0, np.int64)
    s = np.concatenate(ss) if ss else np.zeros(0, np.float32)
    o = np.lexsort((j, i))
    return i[o].astype(np.int64), j[o].astype(np.int64), s[o].astype(np.float32)


def dedup_components(n: int, pi: np.ndarray, pj: np.ndarray) -> np.ndarray:
    """Connected-component label per row = smallest row position in its component (union-find)."""
    parent = np.arange(n, dtype=np.int64)

    def find(a):
        while parent[a] != a:
            parent[a] = parent[parent[a]]
            a = parent[a]
        return a

    for a, b in zip(pi.tolist(), pj.tolist()):
        ra, rb = find(a), find(b)
        if ra != rb:
            if ra < rb:
                parent[rb] = ra
            else:
                parent[ra] = rb
    return np.array([find(a) for a in range(n)], dtype=np.int64)


def dedup_keep_mask(values, x, threshold: float) -> np.ndarray:
    """Boolean keep-mask over rows reproducing sem_dedup.py:45-91 with the lowest-row representative.

    ``values`` are the column values (the reference compares and groups VALUES, so rows with equal values are
    one graph node: none of them is dropped unless the value itself is)."""
    values = list(values)
    n = len(values)
    pi, pj, _ = range_self_join(x, threshold)
    # value-level graph: node = first row position holding that value
    first = {}
    node = np.zeros(n, np.int64)
    for r, v in enumerate(values):
        node[r] = first.setdefault(v, r)
    a, b = node[pi], node[pj]
    diff = a != b  # sem_dedup.py:47,54 drops pairs with equal values
    labels = dedup_components(n, a[diff], b[diff])
    in_pair = np.zeros(n, bool)
    in_pair[a[diff]] = True
    in_pair[b[diff]] = True
    removed_nodes = in_pair & (labels != np.arange(n))
    return ~removed_nodes[node]
