"""Oracle: faiss ``Kmeans(d, k, niter).train(x)`` + ``kmeans.index.search(x, 1)``.  Test infrastructure only.

Follows ``lotus/utils.py:61-65`` and, underneath it, faiss ``Clustering::train`` with the python wrapper's
default ``ClusteringParameters``: nredo=1, spherical=False, seed=1234,
min_points_per_centroid=39, max_points_per_centroid=256, L2 assignment index.

Steps restated: cast to float32 -> subsample ``k*256`` rows by ``rand_perm(n, seed)`` when ``n > k*256`` ->
initial centroids ``x[rand_perm(n', seed+1)[:k]]`` -> ``niter`` x { assign by squared L2 (Flat search, k=1),
objective = sum of distances, centroid = float32 mean in point order, split empty clusters } -> final
assignment of ALL rows to the trained centroids.
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np

from .flat import METRIC_L2, as_f32, flat_search


def _mt19937_raw(seed: int, n: int) -> np.ndarray:
    """n raw draws of std::mt19937(seed) (numpy's legacy RandomState seeds init_genrand identically)."""
    rs = np.random.RandomState(seed & 0xFFFFFFFF)
    return rs._bit_generator.random_raw(n).astype(np.uint32)


def rand_perm(n: int, seed: int, use_c: bool | None = None) -> np.ndarray:
    """faiss ``rand_perm(perm, n, seed)``: Fisher-Yates, ``i2 = i + mt() % (n - i)``."""
    if use_c is not False:
        from . import cbind

        if cbind.available():
            return cbind.rand_perm(n, seed)
        if use_c:
            raise RuntimeError("oracle C library not built")
    perm = np.arange(n, dtype=np.int64)
    if n > 1:
        draws = _mt19937_raw(seed, n - 1).astype(np.int64)
        for i in range(n - 1):
            i2 = i + int(draws[i] % (n - i))
            perm[i], perm[i2] = perm[i2], perm[i]
    return perm


@dataclass
class KMeansResult:
    centroids: np.ndarray  # [k,d] float32
    assign: np.ndarray  # [n] int64, final assignment of all input rows (utils.py:65)
    obj: np.ndarray  # [niter] float32 objective per iteration
    train_ids: np.ndarray  # rows of x used for training (identity when not subsampled)
    nsplit: np.ndarray  # [niter] number of empty clusters re-seeded


def _compute_centroids(x, assign, centroids, use_c):
    from . import cbind

    if use_c is not False and cbind.available():
        return cbind.compute_centroids(x, assign, centroids)
    k = centroids.shape[0]
    sums = np.zeros_like(centroids)
    np.add.at(sums, assign, x)  # unbuffered, in point order == faiss accumulation order
    hassign = np.bincount(assign, minlength=k).astype(np.float32)
    nz = hassign > 0
    centroids[nz] = sums[nz] * (np.float32(1.0) / hassign[nz])[:, None]
    return hassign


def _split_clusters(n, hassign, centroids, use_c):
    from . import cbind

    if use_c is not False and cbind.available():
        return cbind.split_clusters(n, hassign, centroids)
    k, d = centroids.shape
    eps = np.float32(1.0 / 1024.0)
    nsplit = 0
    empties = np.flatnonzero(hassign == 0)
    if len(empties) == 0:
        return 0
    # lazily drawn mt19937(1234) stream
    buf = _mt19937_raw(1234, 4096)
    pos = 0
    sign = np.where(np.arange(d) % 2 == 0, np.float32(1) + eps, np.float32(1) - eps).astype(np.float32)
    sign_o = np.where(np.arange(d) % 2 == 0, np.float32(1) - eps, np.float32(1) + eps).astype(np.float32)
    for ci in empties:
        cj = 0
        while True:
            if pos >= len(buf):
                buf = _mt19937_raw(1234, 2 * len(buf))
            p = np.float32((hassign[cj] - np.float32(1.0)) / np.float32(n - k))
            r = np.float32(np.float32(buf[pos]) / np.float32(4294967295.0))
            pos += 1
            if r < p:
                break
            cj = (cj + 1) % k
        centroids[ci] = centroids[cj]
        centroids[ci] *= sign
        centroids[cj] *= sign_o
        hassign[ci] = np.float32(hassign[cj] / np.float32(2))  # float division as in faiss (float hassign)
        hassign[cj] -= hassign[ci]
        nsplit += 1
    return nsplit


def pair_distances(x, centroids, rows, cols) -> np.ndarray:
    """Squared L2 of (x[rows[i]], centroids[cols[i]]) in the arithmetic of the assignment search above: float32
    ``|x|^2 + |c|^2 - 2 <x, c>`` clamped at 0 (faiss's BLAS path, ``flat._block_better``), the inner product as one float32
    dot per pair."""
    xr = as_f32(x[np.asarray(rows, dtype=np.int64)])
    cr = as_f32(centroids[np.asarray(cols, dtype=np.int64)])
    ip = np.einsum("ij,ij->i", xr, cr, dtype=np.float32)
    from .flat import row_norms_sq

    dis = (row_norms_sq(xr) + row_norms_sq(cr)) - np.float32(2.0) * ip
    return np.maximum(dis, np.float32(0.0)).astype(np.float32)


def flipped_rows(x, centroids, assign_a, assign_b, rel: float = 2e-5):
    """Rows two assignments of ``x`` to ``centroids`` disagree on, and whether every disagreement is a NEAR-TIE: the row's
    distances to its two candidate centroids differ by at most ``rel`` x the larger one in the oracle's float32
    arithmetic (``pair_distances``) - the band inside which a different summation order may pick either.
    -> dict(rows, gaps (relative), near_tie (bool per row), all_near_ties)."""
    a = np.asarray(assign_a, dtype=np.int64).reshape(-1)
    b = np.asarray(assign_b, dtype=np.int64).reshape(-1)
    rows = np.flatnonzero(a != b)
    if len(rows) == 0:
        return {"rows": rows, "gaps": np.zeros(0, np.float32), "near_tie": np.zeros(0, bool), "all_near_ties": True}
    da = pair_distances(x, centroids, rows, a[rows]).astype(np.float64)
    db = pair_distances(x, centroids, rows, b[rows]).astype(np.float64)
    gaps = np.abs(da - db) / np.maximum(np.maximum(da, db), 1e-30)
    near = gaps <= rel
    return {"rows": rows, "gaps": gaps.astype(np.float32), "near_tie": near, "all_near_ties": bool(near.all())}


def kmeans_faiss(x, k: int, niter: int = 20, seed: int = 1234, max_points_per_centroid: int = 256,
                 use_c: bool | None = None, final_assign: bool = True, trace: list | None = None) -> KMeansResult:
    """``trace`` (a list, optional) receives one dict per iteration: ``centroids`` (the centroids the iteration assigns
    against), ``assign`` / ``dist`` (the search's result on the training rows), ``hassign`` (cluster sizes BEFORE the split),
    ``divided`` (centroids after compute_centroids, before split_clusters), ``hassign_after`` (sizes after the split) and
    ``next`` (the centroids the next iteration starts from) - what a step-by-step parity check feeds to the device."""
    x = as_f32(x)
    n, d = x.shape
    if n < k:
        raise ValueError(f"Number of training points ({n}) should be at least as large as number of clusters ({k})")
    train_ids = np.arange(n, dtype=np.int64)
    xt = x
    if n > k * max_points_per_centroid:
        perm = rand_perm(n, seed, use_c)
        train_ids = perm[: k * max_points_per_centroid]
        xt = np.ascontiguousarray(x[train_ids])
    nt = xt.shape[0]
    obj = np.zeros(niter, np.float32)
    nsplits = np.zeros(niter, np.int64)
    if nt == k:
        centroids = xt.copy()  # "n == k: copy points as centroids and stop" (Appendix A.4)
    else:
        perm = rand_perm(nt, seed + 1, use_c)
        centroids = np.ascontiguousarray(xt[perm[:k]]).copy()
        for it in range(niter):
            D, I = flat_search(centroids, xt, 1, METRIC_L2, use_c=use_c)
            obj[it] = np.float32(D[:, 0].sum(dtype=np.float32))
            rec = None
            if trace is not None:
                rec = {"centroids": centroids.copy(), "assign": I[:, 0].copy(), "dist": D[:, 0].copy()}
            hassign = _compute_centroids(xt, I[:, 0], centroids, use_c)
            if rec is not None:
                rec["hassign"], rec["divided"] = hassign.copy(), centroids.copy()
            nsplits[it] = _split_clusters(nt, hassign, centroids, use_c)
            if rec is not None:
                rec["hassign_after"], rec["next"] = hassign.copy(), centroids.copy()
                trace.append(rec)
    assign = np.zeros(0, np.int64)
    if final_assign:
        _, I = flat_search(centroids, x, 1, METRIC_L2, use_c=use_c)
        assign = I[:, 0].copy()
    return KMeansResult(centroids=centroids, assign=assign, obj=obj, train_ids=train_ids, nsplit=nsplits)
