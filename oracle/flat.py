"""Oracle: exact Flat search (faiss ``IndexFlat::search`` semantics).  Test infrastructure only.

Follows, step by step:

* ``lotus/vector_store/faiss_vs.py:23-24`` ``index_factory(d, "Flat", METRIC_INNER_PRODUCT)`` + ``add`` —
  an exact index over C-contiguous float32 rows (faiss python wrapper casts);
* ``faiss_vs.py:75`` ``faiss_index.search(query_vectors, K)`` — best-first ``(D float32 [nq,K], I int64 [nq,K])``,
  inner product descending / squared L2 ascending, tail padded with id -1 and -FLT_MAX / +FLT_MAX when fewer
  than K rows exist (Appendix A.2);
* ``faiss_vs.py:57-72`` the ``ids`` branch — gather the subset, search it, map sub-indices back to the given ids;
* faiss evaluates distances block-wise with one sgemm per (4096 queries x 1024 rows) block and feeds each block
  to a k-best collector (Appendix A.3); L2 on that path is ``|x|^2 + |y|^2 - 2<x,y>`` clamped at 0, and for
  fewer than 20 queries it is the direct ``sum (x-y)^2``.

Ties: faiss only inserts a candidate that is strictly better than the current k-th, so at the boundary the
earlier (lower) id wins; order among equal scores inside the list is implementation-defined there.  The oracle
fixes the total order (score best-first, then id ascending), which coincides with faiss wherever faiss is defined.
"""
from __future__ import annotations

import numpy as np

METRIC_INNER_PRODUCT = 0  # faiss.METRIC_INNER_PRODUCT
METRIC_L2 = 1  # faiss.METRIC_L2

FLT_MAX = np.float32(3.4028234663852886e38)
QUERY_BLOCK = 4096  # faiss distance_compute_blas_query_bs
DB_BLOCK = 1024  # faiss distance_compute_blas_database_bs
BLAS_THRESHOLD = 20  # faiss distance_compute_blas_threshold


def as_f32(x) -> np.ndarray:
    """faiss python wrapper cast (Appendix A.1): C-contiguous float32, 2-D."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    if x.ndim != 2:
        raise ValueError("expected a 2-D array")
    return x


def _ord32(f: np.ndarray) -> np.ndarray:
    """Monotone map float32 -> uint32 (bigger float <=> bigger uint); -0.0 is folded onto +0.0 first."""
    f = (np.asarray(f, dtype=np.float32) + np.float32(0.0)).astype(np.float32)
    u = f.view(np.uint32)
    mask = np.where((u >> np.uint32(31)).astype(bool), np.uint32(0xFFFFFFFF), np.uint32(0x80000000))
    return u ^ mask


def _unord32(u: np.ndarray) -> np.ndarray:
    u = np.asarray(u, dtype=np.uint32)
    mask = np.where((u >> np.uint32(31)).astype(bool), np.uint32(0x80000000), np.uint32(0xFFFFFFFF))
    return (u ^ mask).view(np.float32)


def pack_keys(better: np.ndarray, ids: np.ndarray) -> np.ndarray:
    """(score where larger = better, id) -> uint64 key whose descending order is (score desc, id asc).

    key 0 is reserved for "empty slot" (no real float maps to ord 0 together with id 0xFFFFFFFF).
    """
    hi = _ord32(better).astype(np.uint64) << np.uint64(32)
    lo = (np.uint64(0xFFFFFFFF) - np.asarray(ids, dtype=np.uint64)) & np.uint64(0xFFFFFFFF)
    return hi | lo


def unpack_keys(keys: np.ndarray):
    keys = np.asarray(keys, dtype=np.uint64)
    better = _unord32((keys >> np.uint64(32)).astype(np.uint32))
    ids = (np.uint64(0xFFFFFFFF) - (keys & np.uint64(0xFFFFFFFF))).astype(np.int64)
    empty = keys == 0
    return better, np.where(empty, np.int64(-1), ids), empty


def _block_better(xq_blk, xb_blk, metric, qn_blk, bn_blk, direct_l2):
    """Scores of one block, oriented so that larger = better (IP: the product; L2: minus the squared distance)."""
    if metric == METRIC_INNER_PRODUCT:
        return xq_blk @ xb_blk.T
    if direct_l2:
        diff = xq_blk[:, None, :] - xb_blk[None, :, :]
        return -np.einsum("qbd,qbd->qb", diff, diff, dtype=np.float32)
    ip = xq_blk @ xb_blk.T
    dis = (qn_blk[:, None] + bn_blk[None, :]) - np.float32(2.0) * ip
    np.maximum(dis, np.float32(0.0), out=dis)
    return -dis


def row_norms_sq(x: np.ndarray) -> np.ndarray:
    """|x_i|^2 in float32 (faiss fvec_norms_L2sqr)."""
    return np.einsum("ij,ij->i", x, x, dtype=np.float32).astype(np.float32)


def flat_search(xb, xq, k: int, metric: int = METRIC_INNER_PRODUCT, ids=None, use_c: bool | None = None):
    """Exact top-k of ``xq`` against ``xb`` with faiss Flat semantics.  Returns ``(D float32, I int64)`` [nq,k].

    ``ids`` reproduces ``FaissVS.__call__(..., ids=...)`` (faiss_vs.py:57-72): only rows ``xb[ids]`` take part and
    returned indices are the given ids.
    """
    xb = as_f32(xb)
    xq = as_f32(xq)
    if xq.shape[1] != xb.shape[1]:
        raise ValueError(f"dimension mismatch: queries d={xq.shape[1]} index d={xb.shape[1]}")
    if metric not in (METRIC_INNER_PRODUCT, METRIC_L2):
        raise ValueError("metric must be METRIC_INNER_PRODUCT or METRIC_L2")
    id_map = None
    if ids is not None:
        id_map = np.asarray(ids, dtype=np.int64)
        xb = np.ascontiguousarray(xb[id_map])
    nq, nb = xq.shape[0], xb.shape[0]
    k = int(k)
    if k < 0:
        raise ValueError("k must be >= 0")
    if nb >= 2**32 - 1:
        raise ValueError("oracle supports fewer than 2^32-1 rows")
    keys = np.zeros((nq, k), dtype=np.uint64)
    if k > 0 and nq > 0 and nb > 0:
        direct_l2 = metric == METRIC_L2 and nq < BLAS_THRESHOLD
        qn = row_norms_sq(xq) if metric == METRIC_L2 else None
        bn = row_norms_sq(xb) if metric == METRIC_L2 else None
        c_update = None
        if use_c is not False:
            from . import cbind

            c_update = cbind.topk_update if cbind.available() else None
            if use_c and c_update is None:
                raise RuntimeError("oracle C library not built (run `make -C oracle`)")
        for q0 in range(0, nq, QUERY_BLOCK):
            q1 = min(nq, q0 + QUERY_BLOCK)
            kblk = keys[q0:q1]
            for b0 in range(0, nb, DB_BLOCK):
                b1 = min(nb, b0 + DB_BLOCK)
                better = _block_better(
                    xq[q0:q1], xb[b0:b1], metric, None if qn is None else qn[q0:q1],
                    None if bn is None else bn[b0:b1], direct_l2)
                if c_update is not None:
                    c_update(np.ascontiguousarray(better, dtype=np.float32), b0, kblk)
                else:
                    cand = pack_keys(better, np.arange(b0, b1, dtype=np.int64)[None, :])
                    allk = np.concatenate([kblk, cand], axis=1)
                    if allk.shape[1] > k:
                        part = np.partition(allk, allk.shape[1] - k, axis=1)[:, allk.shape[1] - k:]
                    else:
                        part = allk
                        if part.shape[1] < k:
                            part = np.concatenate(
                                [part, np.zeros((part.shape[0], k - part.shape[1]), np.uint64)], axis=1)
                    kblk[:] = np.sort(part, axis=1)[:, ::-1]
    return _finish(keys, metric, id_map)


def _finish(keys, metric, id_map):
    better, I, empty = unpack_keys(keys)
    if metric == METRIC_INNER_PRODUCT:
        D = np.where(empty, -FLT_MAX, better).astype(np.float32)
    else:
        D = np.where(empty, FLT_MAX, -better).astype(np.float32)
        D = D + np.float32(0.0)  # -(+0) -> +0
    if id_map is not None and I.size:
        safe = np.where(I >= 0, I, 0)
        I = np.where(I >= 0, id_map[safe] if id_map.size else -1, -1)
    return D, I.astype(np.int64)


def flat_search_exact64(xb, xq, k: int, metric: int = METRIC_INNER_PRODUCT):
    """Independent float64 brute force (no blocking, no keys) used to cross-check ``flat_search``.

    Scores are computed in float64 from the float32-cast inputs, rounded to float32 once, then ordered by
    (score best-first, id ascending) with a plain lexsort.
    """
    xb64 = as_f32(xb).astype(np.float64)
    xq64 = as_f32(xq).astype(np.float64)
    nq, nb = xq64.shape[0], xb64.shape[0]
    if metric == METRIC_INNER_PRODUCT:
        s = (xq64 @ xb64.T).astype(np.float32)
        order_key = -s.astype(np.float64)
    else:
        d2 = ((xq64[:, None, :] - xb64[None, :, :]) ** 2).sum(-1) if nq * nb * xb64.shape[1] < 5e7 else (
            (xq64**2).sum(1)[:, None] + (xb64**2).sum(1)[None, :] - 2 * xq64 @ xb64.T)
        s = np.maximum(d2, 0).astype(np.float32)
        order_key = s.astype(np.float64)
    D = np.full((nq, k), -FLT_MAX if metric == METRIC_INNER_PRODUCT else FLT_MAX, np.float32)
    I = np.full((nq, k), -1, np.int64)
    ids = np.arange(nb)
    for q in range(nq):
        o = np.lexsort((ids, order_key[q]))[:k]
        D[q, : len(o)] = s[q, o]
        I[q, : len(o)] = o
    return D, I
