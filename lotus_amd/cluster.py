"""k-means behind ``sem_cluster_by`` - MI355X-native replacement for ``lotus.utils.cluster`` (``lotus/utils.py:14-72``).

The reference hard-codes ``faiss.Kmeans(d, k, niter=niter).train(vec_set)`` followed by ``kmeans.index.search(vec_set, 1)``
(``utils.py:61-65``) instead of going through the ``VS`` plugin, so a drop-in needs this module:

* :func:`kmeans` - faiss-parity Lloyd iterations on the GPU: subsample ``k*256`` rows with faiss's ``rand_perm`` when
  there are more, initial centroids ``x[rand_perm(n', seed+1)[:k]]``, per iteration {assign by squared L2 = the tile
  kernel in top-1 mode, deterministic in-row-order centroid sums, faiss's empty-cluster split}, then the final
  assignment of all rows.  ``max_points_per_centroid=None`` trains on all rows (BASELINE
  configs[4] "full-data" mode).
* :func:`cluster` - same signature and checks as ``lotus.utils.cluster``.
* :func:`install` - monkey-patches ``lotus.utils.cluster`` (the accessors look it up at call time,
  ``sem_cluster_by.py:74``).
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np

from . import _capi


@dataclass
class KMeansResult:
    centroids: np.ndarray  # [k,d] float32
    assign: np.ndarray  # [n] int64
    obj: np.ndarray  # [niter] float32, sum of squared distances per iteration
    nsplit: np.ndarray  # [niter] empty clusters re-seeded
    train_ids: np.ndarray  # rows used for training


def _dist_ctx(shard: bool, pg=None):
    if not shard:
        return None, 0, 1
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()):
        return None, 0, 1
    return dist, dist.get_rank(pg), dist.get_world_size(pg)


def kmeans(x, k: int, niter: int = 20, seed: int = 1234, max_points_per_centroid: int | None = 256, backend=None,
           pack_mode: int | None = None, packed=None, shard: bool = False, process_group=None,
           final_assign: bool = True, centroid_precision: str = "fp32") -> KMeansResult:
    """faiss-parity k-means.  ``x``: host matrix [n,d] (float16/32/64); ``packed`` optionally its device image.

    ``centroid_precision="fp32"`` (default) keeps the centroids fp32-accurate on the device (fp16 hi|lo pair) as
    faiss does, whatever the storage of the points; ``"fp16"`` rounds them to fp16 (half the MFMA work when the
    points are fp16, ~1e-3 relative distance error)."""
    if backend is None:
        from .backend import HipBackend

        backend = HipBackend()
    be = backend
    x = np.asarray(x)
    if x.ndim != 2:
        raise ValueError("x must be 2-D")
    n, d = x.shape
    k = int(k)
    if n < k:
        raise ValueError(f"Number of training points ({n}) should be at least as large as number of clusters ({k})")
    if pack_mode is None:
        pack_mode = _capi.PACK_F16 if x.dtype == np.float16 else _capi.PACK_SPLIT
    if packed is None:
        packed = be.pack(x, pack_mode)
    if centroid_precision not in ("fp32", "fp16"):
        raise ValueError("centroid_precision must be 'fp32' or 'fp16'")
    cmode = _capi.PACK_SPLIT if centroid_precision == "fp32" else pack_mode
    dist, rank, world = _dist_ctx(shard, process_group)

    train_ids = np.arange(n, dtype=np.int64)
    if max_points_per_centroid is not None and n > k * max_points_per_centroid:
        train_ids = be.rand_perm(n, seed)[: k * max_points_per_centroid]
    nt = len(train_ids)
    obj = np.zeros(niter, np.float32)
    nsplit = np.zeros(niter, np.int64)
    x32 = None

    def rows32(idx):
        return np.ascontiguousarray(x[idx], dtype=np.float32)

    if nt == k:
        centroids = rows32(train_ids)  # faiss: "n == k: copy points as centroids and stop"
    else:
        perm = be.rand_perm(nt, seed + 1)
        centroids = rows32(train_ids[perm[:k]])
        # this rank's share of the training rows (all of them without sharding)
        per = -(-nt // world)
        lo, hi = min(nt, rank * per), min(nt, (rank + 1) * per)
        local_ids = train_ids[lo:hi]
        if nt == n and world == 1:
            train = packed
        else:
            train = be.gather(packed, be.to_device(local_ids))
        for it in range(niter):
            cpk = be.pack(centroids, cmode)
            keys = be.search_keys(cpk, train, 1, _capi.METRIC_L2)
            D, I = be.keys_to_result(keys, _capi.METRIC_L2)
            sums, counts = be.kmeans_accumulate(train, I.reshape(-1), k)
            o = D.sum()
            if dist is not None and world > 1:
                dist.all_reduce(sums, group=process_group)
                dist.all_reduce(counts, group=process_group)
                dist.all_reduce(o, group=process_group)
            obj[it] = float(o.item())
            hs = counts.cpu().numpy().astype(np.float32)
            sm = sums.cpu().numpy()
            nz = hs > 0
            centroids = np.ascontiguousarray(centroids, dtype=np.float32)
            centroids[nz] = sm[nz] * (np.float32(1.0) / hs[nz])[:, None]  # faiss: c *= 1 / count
            nsplit[it] = be.split_clusters(nt, hs, centroids)
    assign = np.zeros(0, np.int64)
    if final_assign:
        cpk = be.pack(centroids, cmode)
        keys = be.search_keys(cpk, packed, 1, _capi.METRIC_L2)
        _, I = be.keys_to_result(keys, _capi.METRIC_L2)
        assign = I.reshape(-1).cpu().numpy().astype(np.int64)
    return KMeansResult(centroids=np.asarray(centroids, np.float32), assign=assign, obj=obj, nsplit=nsplit,
                        train_ids=train_ids)


def cluster(col_name: str, ncentroids: int):
    """Drop-in for ``lotus.utils.cluster``: returns ``ret(df, niter=20, verbose=False, method="kmeans")`` giving the
    cluster id of every row (``lotus/utils.py:26-70``; same checks, same error messages)."""

    def ret(df, niter: int = 20, verbose: bool = False, method: str = "kmeans"):
        import lotus  # the accessor layer this plugs into

        if col_name not in df.columns:
            raise ValueError(f"Column {col_name} not found in DataFrame")
        if ncentroids > len(df):
            raise ValueError(
                f"Number of centroids must be less than number of documents. {ncentroids} > {len(df)}")
        rm = lotus.settings.rm
        vs = lotus.settings.vs
        if vs is not None and not hasattr(vs, "packed_rows") and _ORIGINAL is not None:
            # another vector store is configured (e.g. FaissVS): leave its k-means to the reference implementation
            return _ORIGINAL(col_name, ncentroids)(df, niter, verbose, method)
        if rm is None or vs is None:
            raise ValueError(
                "The retrieval model must be an instance of RM, and the vector store must be an instance of VS. "
                "Please configure a valid retrieval model using lotus.settings.configure()")
        try:
            col_index_dir = df.attrs["index_dirs"][col_name]
        except KeyError:
            raise ValueError(f"Index directory for column {col_name} not found in DataFrame")
        if vs.index_dir != col_index_dir:
            vs.load_index(col_index_dir)
        assert vs.index_dir == col_index_dir
        ids = df.index.tolist()
        vec_set = vs.get_vectors_from_index(col_index_dir, ids)
        backend = getattr(vs, "backend", None)
        packed = None
        if hasattr(vs, "packed_rows"):
            try:
                packed = vs.packed_rows(ids)  # reuse the resident device image instead of re-uploading vec_set
            except ValueError:
                packed = None
        res = kmeans(vec_set, ncentroids, niter=niter, backend=backend, packed=packed,
                     pack_mode=None if packed is None else packed.mode)
        if verbose:
            for it, o in enumerate(res.obj):
                print(f"  Iteration {it} objective={o:.6g} splits={int(res.nsplit[it])}")
        return res.assign

    return ret


_ORIGINAL = None


def install() -> None:
    """Route ``lotus.utils.cluster`` (used by ``sem_cluster_by`` and as a ``sem_partition_by`` partition function)
    through the GPU k-means.  Idempotent; :func:`uninstall` restores the reference function."""
    global _ORIGINAL
    import lotus.utils

    if _ORIGINAL is None:
        _ORIGINAL = lotus.utils.cluster
    lotus.utils.cluster = cluster


def uninstall() -> None:
    global _ORIGINAL
    if _ORIGINAL is not None:
        import lotus.utils

        lotus.utils.cluster = _ORIGINAL
        _ORIGINAL = None
