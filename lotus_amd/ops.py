"""Fast-path accessors: the reference operators' semantics without their Python hot loops.

``sem_sim_join`` in the reference walks Q x K results in a Python double loop with set look-ups and builds a list of
tuples (``lotus/sem_ops/sem_sim_join.py:132-145``), and ``sem_dedup`` materialises an N x N join
(``sem_dedup.py:45``).  These functions produce the same frames from the same inputs with vectorised post-processing,
and work with any ``VS`` that returns ndarrays (they are plain functions: ``sem_sim_join(df1, df2, ...)``)."""
from __future__ import annotations

import numpy as np
import pandas as pd

from . import dedup as _dedup


def _is_vectors(x) -> bool:
    """ndarray or device tensor of ready-made vectors (``rm.py:77-78`` passes ndarrays through; so do we for tensors)."""
    return isinstance(x, np.ndarray) or (hasattr(x, "is_cuda") and hasattr(x, "data_ptr"))


def _settings(rm, vs):
    if rm is None or vs is None:
        try:
            import lotus

            rm = rm if rm is not None else lotus.settings.rm
            vs = vs if vs is not None else lotus.settings.vs
        except Exception:
            pass
    if vs is None:
        raise ValueError("a vector store is required: pass vs= or configure lotus.settings")
    return rm, vs


def sem_index(df: pd.DataFrame, col_name: str, index_dir: str, rm=None, vs=None, embeddings=None) -> pd.DataFrame:
    """``df.sem_index`` (``sem_index.py:61-77``): embed the column (or take ``embeddings``), index it, remember the dir."""
    rm, vs = _settings(rm, vs)
    if embeddings is None:
        if rm is None:
            raise ValueError("a retrieval model (rm) or precomputed embeddings are required")
        embeddings = rm(df[col_name].tolist())
    vs.index(df[col_name], embeddings, index_dir)
    df.attrs.setdefault("index_dirs", {})[col_name] = index_dir
    return df


def sem_search(df: pd.DataFrame, col_name: str, query, K: int, rm=None, vs=None, return_scores: bool = False,
               suffix: str = "_sim_score") -> pd.DataFrame:
    """``df.sem_search`` without the K-doubling loop (``sem_search.py:116-144``): rows that are no longer in ``df``
    are excluded up front by passing their positions as ``ids``, so one search suffices."""
    rm, vs = _settings(rm, vs)
    col_index_dir = df.attrs["index_dirs"][col_name]
    if vs.index_dir != col_index_dir:
        vs.load_index(col_index_dir)
    K = min(int(K), len(df))
    qv = query if _is_vectors(query) else rm.convert_query_to_query_vector(query)
    live = np.asarray(df.index)
    if K >= len(df) > 0 and hasattr(vs, "scores") and getattr(vs, "metric", 0) == 0:
        # every live row is wanted (the cascade callers: sem_filter.py:491-497 proxy scores, sem_topk.py:786-788): one
        # score row without any top-k machinery, ordered on the host exactly as the search would (score best-first,
        # lower row id first among equals) - SURVEY.md 8(f).4
        sc_all = vs.scores(qv, ids=live.tolist())[0]
        order = np.lexsort((live, -sc_all.astype(np.float64)))
        idx, sc = live[order], sc_all[order]
    else:
        out = vs(qv, K, ids=live.tolist())
        idx = np.asarray(out.indices)[0]
        sc = np.asarray(out.distances)[0]
    ok = idx >= 0
    new_df = df.loc[idx[ok]]
    new_df.attrs["index_dirs"] = df.attrs.get("index_dirs", None)
    if return_scores:
        new_df["vec_scores" + suffix] = sc[ok]
    return new_df


def sem_sim_join(df1: pd.DataFrame, df2: pd.DataFrame, left_on: str, right_on: str, K: int, rm=None, vs=None,
                 lsuffix: str = "", rsuffix: str = "", score_suffix: str = "", keep_index: bool = False) -> pd.DataFrame:
    """``df1.sem_sim_join(df2, ...)`` (``sem_sim_join.py:84-166``) with the post-filter and the result frame built from
    arrays.  Same columns, same row order (left rows in order, best match first), same ``_scores`` dtype."""
    rm, vs = _settings(rm, vs)
    if isinstance(df2, pd.Series):
        if df2.name is None:
            raise ValueError("Other Series must have a name")
        df2 = pd.DataFrame({df2.name: df2})
    if left_on in df1.attrs.get("index_dirs", []):
        qdir = df1.attrs["index_dirs"][left_on]
        if vs.index_dir != qdir:
            vs.load_index(qdir)
        try:
            queries = vs.get_vectors_from_index(qdir, df1.index)
        except NotImplementedError:
            queries = df1[left_on]
    else:
        queries = df1[left_on]
    try:
        col_index_dir = df2.attrs["index_dirs"][right_on]
    except KeyError:
        raise ValueError(f"Index directory for column {right_on} not found in DataFrame")
    if vs.index_dir != col_index_dir:
        vs.load_index(col_index_dir)
    qv = queries if _is_vectors(queries) else rm.convert_query_to_query_vector(queries)
    right_index = np.asarray(df2.index)
    # the reference hands a Python list over (sem_sim_join.py:132-134); HipVS takes the array as it is (a million-entry
    # list costs ~50 ms to build and as much to turn back into an array)
    native = hasattr(vs, "packed_rows") and right_index.dtype.kind in "iu"
    out = vs(qv, K, ids=right_index if native else right_index.tolist())
    I = np.asarray(out.indices)
    D = np.asarray(out.distances)
    if I.size and int(I.min()) >= 0:
        # no padded slot (the usual case: K <= rows of df2): views instead of three masked copies of a million entries
        qpos = np.repeat(np.arange(I.shape[0]), I.shape[1])
        right_ids, scores = I.reshape(-1), D.reshape(-1)
    else:
        ok = I >= 0  # ids come from df2.index, so every non-padded hit is a right row
        qpos = np.broadcast_to(np.arange(I.shape[0])[:, None], I.shape)[ok]
        right_ids, scores = I[ok], D[ok]
    left_ids = np.asarray(df1.index)[qpos]
    fast = _joined_frame(df1, df2, qpos, left_ids, right_ids, scores, lsuffix, rsuffix, score_suffix, keep_index)
    if fast is not None:
        return fast
    d1 = df1.copy()
    d2 = df2.copy()
    d1["_left_id"] = d1.index
    d2["_right_id"] = d2.index
    temp = pd.DataFrame({"_left_id": left_ids, "_right_id": right_ids, "_scores" + score_suffix: scores})
    joined = d1.join(temp.set_index("_left_id"), how="right", on="_left_id").join(
        d2.set_index("_right_id"), how="left", on="_right_id", lsuffix=lsuffix, rsuffix=rsuffix)
    if not keep_index:
        joined.drop(columns=["_left_id", "_right_id"], inplace=True)
    return joined


def _joined_frame(df1, df2, qpos, left_ids, right_ids, scores, lsuffix, rsuffix, score_suffix, keep_index):
    """The frame the reference's two joins (``sem_sim_join.py:152-162``) produce, built from positional takes: rows =
    one per (left row, match) in the order given, index = the left row's label, columns = df1's, ``_left_id``,
    ``_right_id``, ``_scores``, df2's (overlapping names suffixed as ``DataFrame.join`` does).  Returns ``None`` when
    the shortcut does not apply (duplicate labels, non-default column index) - the caller then runs the joins."""
    if not (df1.index.is_unique and df2.index.is_unique and df1.columns.is_unique and df2.columns.is_unique):
        return None
    if df1.columns.nlevels != 1 or df2.columns.nlevels != 1 or df1.index.nlevels != 1 or df2.index.nlevels != 1:
        return None
    mid = ["_left_id", "_right_id", "_scores" + score_suffix]
    lcols = list(df1.columns) + mid
    rcols = list(df2.columns)
    if any(c in df1.columns for c in mid) or "_right_id" in df2.columns:
        return None  # the joins give these their own (version-dependent) treatment
    overlap = set(lcols) & set(rcols)
    if overlap and not lsuffix and not rsuffix:
        return None  # DataFrame.join raises here; let it
    if isinstance(df2.index, pd.RangeIndex) and df2.index.start == 0 and df2.index.step == 1:
        rpos = np.asarray(right_ids)  # labels ARE positions (a frame that was never re-indexed): no hash look-up
        if rpos.size and (int(rpos.min()) < 0 or int(rpos.max()) >= len(df2)):
            return None
    else:
        rpos = df2.index.get_indexer(right_ids)
        if (rpos < 0).any():
            return None
    index = df1.index.take(qpos)
    left = df1.take(qpos)
    right = df2.take(rpos)
    left.index = index
    right.index = index
    ren_l = {c: f"{c}{lsuffix}" for c in overlap if c in df1.columns}
    ren_r = {c: f"{c}{rsuffix}" for c in overlap}
    if ren_l:
        left = left.rename(columns=ren_l)
    if ren_r:
        right = right.rename(columns=ren_r)
    mid_names = [f"{c}{lsuffix}" if c in overlap else c for c in mid]
    middle = pd.DataFrame({mid_names[0]: np.asarray(left_ids), mid_names[1]: np.asarray(right_ids),
                           mid_names[2]: scores}, index=index)
    if len(set(left.columns) | set(middle.columns) | set(right.columns)) != left.shape[1] + 3 + right.shape[1]:
        return None  # suffixed names collide with existing ones: leave it to pandas
    if not keep_index:
        middle = middle[[mid_names[2]]]
    return pd.concat([left, middle, right], axis=1, copy=False)


def sem_dedup(df: pd.DataFrame, col_name: str, threshold: float, rm=None, vs=None, shard: bool = False) -> pd.DataFrame:
    """``df.sem_dedup`` (``sem_dedup.py:32-91``) through the GPU threshold self-join: O(pairs) memory instead of
    O(N^2).  Survivor of each duplicate group = its first value in frame order.

    ``threshold`` is compared with the similarity score exactly as the reference does (``_scores > threshold``,
    ``sem_dedup.py:46``).  That only means something for a similarity metric: with an L2 vector store the reference's
    ``_scores`` are squared distances (larger = farther) and its dedup keeps the wrong side; here an L2 store is
    refused instead of silently thresholding the negated distance."""
    rm, vs = _settings(rm, vs)
    if getattr(vs, "metric", 0) != 0:
        raise ValueError("sem_dedup thresholds a similarity: configure the vector store with METRIC_INNER_PRODUCT")
    col_index_dir = df.attrs["index_dirs"][col_name]
    if vs.index_dir != col_index_dir:
        vs.load_index(col_index_dir)
    packed = vs.packed_rows(df.index)
    i, j, _ = _dedup.threshold_pairs(vs.backend, packed, threshold, vs.metric, shard=shard)
    mask = _dedup.keep_mask(df[col_name].tolist(), i, j)
    return df[mask]


def sem_cluster_by(df: pd.DataFrame, col_name: str, ncentroids: int, niter: int = 20, verbose: bool = False,
                   rm=None, vs=None) -> pd.DataFrame:
    """``df.sem_cluster_by`` (``sem_cluster_by.py:57-86``): adds the ``cluster_id`` column computed by the GPU k-means
    (same checks as ``lotus.utils.cluster``)."""
    from .cluster import kmeans

    rm, vs = _settings(rm, vs)
    if col_name not in df.columns:
        raise ValueError(f"Column {col_name} not found in DataFrame")
    if ncentroids > len(df):
        raise ValueError(f"Number of centroids must be less than number of documents. {ncentroids} > {len(df)}")
    try:
        col_index_dir = df.attrs["index_dirs"][col_name]
    except KeyError:
        raise ValueError(f"Index directory for column {col_name} not found in DataFrame")
    if vs.index_dir != col_index_dir:
        vs.load_index(col_index_dir)
    ids = df.index.tolist()
    if hasattr(vs, "kmeans"):
        res = vs.kmeans(None, ncentroids, niter=niter, ids=ids, return_result=True)
    else:
        res = kmeans(vs.get_vectors_from_index(col_index_dir, ids), ncentroids, niter=niter,
                     backend=getattr(vs, "backend", None))
    if verbose:
        for it, o in enumerate(res.obj):
            print(f"  Iteration {it} objective={o:.6g}")
    out = df.copy()
    out["cluster_id"] = pd.Series(res.assign, index=df.index)
    return out
