"""Scalable ``sem_dedup`` arithmetic: threshold self-join on the GPU + connected components on the host.

The reference computes ``sem_sim_join(self, K=len(df))`` - N^2 scores, N^2 ids - and only then filters
``_scores > threshold`` (``lotus/sem_ops/sem_dedup.py:45-46``); it cannot run beyond a few tens of thousands of rows.
Here the tile kernel emits just the qualifying pairs ``i < j`` (``lvs_range_join``, strict ``>`` as the reference),
after which the reference's own rule is applied unchanged: pairs whose VALUES differ form an undirected graph over
values, and every value of a connected component except one is removed (``sem_dedup.py:47-91``).  The reference picks
the survivor from a hash-ordered ``set``; we pick the value that appears first in the frame (documented deviation,
DESIGN.md)."""
from __future__ import annotations

import numpy as np

from . import _capi


def threshold_pairs(backend, packed, threshold: float, metric: int = _capi.METRIC_IP, shard: bool = False,
                    process_group=None):
    """-> (i, j, score) numpy arrays of all row pairs i < j of ``packed`` with score > threshold, sorted by (i, j).

    With ``shard=True`` and ``torch.distributed`` initialised the rows are replicated on every rank, 256-row query
    tiles are dealt round-robin to the ranks, and the per-rank pair lists are exchanged with an all-gather."""
    rank, world, dist = 0, 1, None
    if shard:
        import torch.distributed as dist_mod

        if dist_mod.is_available() and dist_mod.is_initialized():
            dist = dist_mod
            rank, world = dist.get_rank(process_group), dist.get_world_size(process_group)
    q, j, s = backend.range_join(packed, packed, threshold, metric, q_row0=0, stride=world, phase=rank)
    q, j, s = q.cpu().numpy(), j.cpu().numpy(), s.cpu().numpy()
    if dist is not None and world > 1:
        parts = [None] * world
        dist.all_gather_object(parts, (q, j, s), group=process_group)  # variable-length lists, host side
        q = np.concatenate([p[0] for p in parts])
        j = np.concatenate([p[1] for p in parts])
        s = np.concatenate([p[2] for p in parts])
    order = np.lexsort((j, q))
    return q[order].astype(np.int64), j[order].astype(np.int64), s[order].astype(np.float32)


def component_labels(n: int, i: np.ndarray, j: np.ndarray) -> np.ndarray:
    """Connected-component label of every node = smallest node id in its component."""
    if len(i) == 0:
        return np.arange(n, dtype=np.int64)
    from scipy.sparse import coo_matrix
    from scipy.sparse.csgraph import connected_components

    g = coo_matrix((np.ones(len(i), np.int8), (i, j)), shape=(n, n))
    _, lab = connected_components(g, directed=False)
    first = np.full(lab.max() + 1, n, dtype=np.int64)
    np.minimum.at(first, lab, np.arange(n, dtype=np.int64))
    return first[lab]


def keep_mask(values, i: np.ndarray, j: np.ndarray) -> np.ndarray:
    """Rows kept by ``sem_dedup`` given the qualifying row pairs (positions into ``values``).

    Value-level semantics of ``sem_dedup.py:47-91``: rows holding equal values are one node (pairs between them are
    ignored), and a value is dropped iff it is in a component of size > 1 and is not that component's first value."""
    values = list(values)
    n = len(values)
    first_row: dict = {}
    node = np.empty(n, np.int64)
    for r, v in enumerate(values):
        node[r] = first_row.setdefault(v, r)
    a, b = node[np.asarray(i, np.int64)], node[np.asarray(j, np.int64)]
    diff = a != b
    a, b = a[diff], b[diff]
    labels = component_labels(n, a, b)
    in_pair = np.zeros(n, bool)
    in_pair[a] = True
    in_pair[b] = True
    removed = in_pair & (labels != np.arange(n))
    return ~removed[node]
