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

from . import _capi, _dist

# row ranges of an exhaustive iteration from 2^21 training rows on (an int, or a tuple of fractions).  Nothing hides the sums of
# the LAST range, so it is the shortest: 10 M x 1 024 x 768, same box, ms per iteration (profiles/r06c_km_parts_probe.log):
# one range 21.5, 3 / 4 / 6 equal ranges 20.5 / 20.2-20.3 / 20.5, 30 / 30 / 25 / 15 % 19.8
PARTS_DEFAULT = (0.30, 0.30, 0.25, 0.15)


def range_cuts(n: int, fracs) -> list[int]:
    """Row ranges [cuts[i], cuts[i + 1]) of ``n`` rows in the proportions ``fracs``: cut points on multiples of 4 096 rows (whole
    tiles of every kernel involved), the last range takes the remainder."""
    total = float(sum(fracs))
    acc, cuts = 0.0, [0]
    for f in list(fracs)[:-1]:
        acc += float(f) / total
        cuts.append(max(cuts[-1], min(n, int(n * acc) // 4096 * 4096)))
    cuts.append(n)
    return cuts
SIDE_STREAM_PRIORITY = 0    # of the stream the sums of a range run on
# the certificate's tail of a range (count read-back, pair dot products, exact search of the open rows) also runs on the side
# stream, under the next range's search - the main stream then goes from one assignment launch straight to the next
USE_ITERATION_OP = True  # single-range exhaustive iterations through lvs_kmeans_iteration (one C-ABI call each)
PIPELINE_CERTIFICATES = True


@dataclass
class KMeansResult:
    centroids: np.ndarray  # [k,d] float32
    assign: np.ndarray  # [n] int64
    obj: np.ndarray  # [niter] float32, sum of squared distances per iteration
    nsplit: np.ndarray  # [niter] empty clusters re-seeded
    train_ids: np.ndarray  # rows used for training


def kmeans(x, k: int, niter: int = 20, seed: int = 1234, max_points_per_centroid: int | None = 256, backend=None,
           pack_mode: int | None = None, packed=None, shard: bool = False, process_group=None,
           final_assign: bool = True, centroid_precision: str = "fp32", n_total: int | None = None,
           local_pos=None, stats: dict | None = None, bounds: bool | None = None,
           trace: list | None = None, parts: int | None = None) -> KMeansResult:
    """faiss-parity k-means (``faiss.Kmeans(d, k, niter).train(x)`` + ``index.search(x, 1)``, ``lotus/utils.py:61-65``).

    ``x``: host matrix [n,d] (float16/32/64) and/or ``packed``: its device image.  Everything after the packing runs
    on the device image: initial centroids are unpacked from it, assignment is the tile kernel in top-1 / squared-L2
    mode (certified one-pass form for fp32-accurate centroids), sums are accumulated in row order, and the objective,
    the centroid division, faiss's empty-cluster split (its ``std::mt19937`` replayed by a device thread) and the
    repacking of the centroids happen on the device too: an iteration is a chain of launches whose only host round trip
    is the count of uncertified rows inside ``nearest``; objectives and split counts are read once, after the loop.
    ``stats`` (dict) collects ``uncertified`` / ``queries`` of the certified assignments (and ``searched_rows`` per iteration).

    ``bounds`` (default: on from 2^20 training rows with fp32-accurate centroids): exact distance bounds across iterations
    (Hamerly 2010) - a row whose upper bound to its centroid stays below its lower bound to every other centroid after the
    centroids moved keeps its assignment without a search (``lvs_kmeans_bounds_step``); all other rows are searched and get
    fresh bounds.  Same assignments, sums, centroids and objectives as the exhaustive iteration - the sums still run over
    all rows in row order - but once the centroids settle an iteration costs little more than that one pass over the rows.

    Multi-GPU (``shard=True`` with ``torch.distributed`` initialised):
      * rows replicated (``packed`` holds all ``n`` rows on every rank): the training rows are dealt to the ranks in
        contiguous slices of the training order, the final assignment in contiguous row slices;
      * rows sharded (``packed`` holds the rows at positions ``local_pos`` - ascending - of an ``n_total``-row matrix,
        e.g. this rank's shard of a row-sharded ``HipVS``): every rank trains on the training rows it holds and
        assigns the rows it holds.
      Either way one all-reduce of the ``[k,d]`` sums, ``[k]`` counts and the objective per iteration, and one
      all-gather of the final cluster ids; nothing else crosses the ranks.

    ``parts`` (an int or a sequence of fractions; default: ``PARTS_DEFAULT`` from 2^21 training rows on, exhaustive iterations
    only): the training rows go through an iteration in that many consecutive ranges, and the in-row-order sums of one range run on a side stream UNDER the assignment search of the
    next (the sums kernel needs 82 VGPRs: one of its waves fits on a SIMD beside the two waves of the assignment kernel).  The
    sums continue across the ranges in row order, so results are bit-identical to ``parts=1``; at 10 M x 1 024 x 768 an iteration
    drops from 21.5 to 19.8 ms (tools/km_parts_probe.py; search + sums alone 21.0 -> 19.2 ms, tools/overlap_probe.py).

    ``trace`` (a list, optional; parity tooling): one dict per iteration with device copies of the centroids the iteration
    assigned against (``centroids``, float32 [k,d], the rows' scaled domain) and of the assignment's result keys (``keys``).

    ``centroid_precision="fp32"`` (default) keeps the centroids fp32-accurate on the device (fp16 hi|lo pair) as
    faiss does, whatever the storage of the points; ``"fp16"`` rounds them to fp16 (half the MFMA work when the
    points are fp16, ~1e-3 relative distance error)."""
    if backend is None:
        from .backend import HipBackend

        backend = HipBackend()
    be = backend
    if packed is None:
        x = np.asarray(x)
        if x.ndim != 2:
            raise ValueError("x must be 2-D")
        if pack_mode is None:
            pack_mode = _capi.PACK_F16 if x.dtype == np.float16 else _capi.PACK_SPLIT
        packed = be.pack(x, pack_mode, exp="auto", check=True)
    d = packed.d
    pexp = int(getattr(packed, "exp", 0))  # the device rows hold x * 2^pexp: centroids, sums and objectives below live in
    # that scaled domain (power-of-two scaling commutes with every float32 operation of the loop) and are unscaled at the end
    n = int(n_total) if n_total is not None else packed.n
    k = int(k)
    if n < k:
        raise ValueError(f"Number of training points ({n}) should be at least as large as number of clusters ({k})")
    if centroid_precision not in ("fp32", "fp16"):
        raise ValueError("centroid_precision must be 'fp32' or 'fp16'")
    cmode = _capi.PACK_SPLIT if centroid_precision == "fp32" else packed.mode
    dist, rank, world = _dist.context(shard, process_group)
    sharded_rows = packed.n != n  # this rank holds only the rows at local_pos
    if sharded_rows:
        if local_pos is None:
            raise ValueError("a partial device image needs local_pos (the positions of its rows)")
        local_pos = np.asarray(local_pos, dtype=np.int64)
        if len(local_pos) != packed.n:
            raise ValueError("local_pos must name every row of the device image")
        if dist is None and packed.n != n:
            raise ValueError("a partial device image needs shard=True and an initialised process group")

    def local_rows(ids: np.ndarray) -> tuple[np.ndarray, np.ndarray]:
        """(mask of `ids` held by this rank, their row numbers in `packed`)."""
        if not sharded_rows:
            return np.ones(len(ids), bool), ids
        at = np.searchsorted(local_pos, ids)
        at[at >= len(local_pos)] = 0
        held = local_pos[at] == ids if len(local_pos) else np.zeros(len(ids), bool)
        return held, at[held]

    train_ids = np.arange(n, dtype=np.int64)
    if max_points_per_centroid is not None and n > k * max_points_per_centroid:
        train_ids = be.rand_perm(n, seed, k * max_points_per_centroid)  # only the prefix is used: O(k * 256) host work
    nt = len(train_ids)
    obj = np.zeros(niter, np.float32)
    nsplit = np.zeros(niter, np.int64)

    def centroid_rows(ids: np.ndarray):
        """float32 [len(ids), d] values of the rows `ids` on the device (every rank gets all of them)."""
        held, rows = local_rows(ids)
        if not sharded_rows:
            return be.unpack(packed, be.to_device(rows), raw=True)
        import torch

        vals = torch.zeros((len(ids), d), dtype=torch.float32, device=packed.rows.device)
        if held.any():
            vals[be.to_device(np.flatnonzero(held))] = be.unpack(packed, be.to_device(rows), raw=True)
        _dist.all_reduce_sum_([vals], process_group)  # every row is held by exactly one rank: x + 0 + ... is exact
        return vals

    if nt == k:
        centroids = centroid_rows(train_ids)  # faiss: "n == k: copy points as centroids and stop"
    else:
        perm = be.rand_perm(nt, seed + 1, k)
        centroids = centroid_rows(train_ids[perm[:k]])
        # the training rows this rank works on
        if sharded_rows:
            held, rows = local_rows(train_ids)
            train = be.gather(packed, be.to_device(rows))
        elif dist is not None:
            per = -(-nt // world)
            lo, hi = min(nt, rank * per), min(nt, (rank + 1) * per)
            train = be.gather(packed, be.to_device(train_ids[lo:hi]))
        elif nt == n:
            train = packed
        else:
            train = be.gather(packed, be.to_device(train_ids))
        import torch

        dev = train.rows.device
        x2 = train.norms.double().sum().reshape(1)  # sum of |x_i|^2 over this rank's training rows (constant over the iterations)
        obj_dev = torch.zeros((max(niter, 1),), dtype=torch.float64, device=dev)
        nsplit_dev = torch.zeros((max(niter, 1),), dtype=torch.int32, device=dev)
        cpk, cstats = be.kmeans_pack_centroids(centroids, cmode, exp=pexp)
        use_bounds = bounds if bounds is not None else (train.n >= (1 << 20))
        use_bounds = bool(use_bounds) and cmode == _capi.PACK_SPLIT and hasattr(be, "kmeans_bounds_step") and k >= 2
        if use_bounds:
            b_assign = torch.full((train.n,), -1, dtype=torch.int32, device=dev)
            b_ub = torch.zeros((train.n,), dtype=torch.float32, device=dev)
            b_lb = torch.zeros((train.n,), dtype=torch.float32, device=dev)
            keys = None
        # stats["time_iterations"] = True: device time of every iteration from events on the launch stream -> stats["iteration_ms"]
        timed = stats is not None and bool(stats.get("time_iterations")) and dev.type == "cuda"
        marks = []
        if parts is None:
            parts = PARTS_DEFAULT if train.n >= (1 << 21) else 1
        fracs = [1.0 / int(parts)] * int(parts) if isinstance(parts, int) else [float(f) for f in parts]
        can_overlap = not use_bounds and hasattr(be, "kmeans_accumulate_keys_into") and dev.type == "cuda"
        nparts = len(fracs) if (can_overlap and min(fracs) * train.n >= 65536) else 1
        pipelined = nparts > 1 and PIPELINE_CERTIFICATES and hasattr(be, "nearest_begin")
        # one range, no distance bounds, fp32-accurate centroids of at most 16 384: the iteration is the ABI's single call
        use_iter_op = (USE_ITERATION_OP and nparts == 1 and not use_bounds and cmode == _capi.PACK_SPLIT and hasattr(be, "kmeans_iteration")
                       and k <= _capi.NEAREST3_MAX_ROWS and dev.type == "cuda" and niter > 0)
        if use_iter_op:
            iter_keys = torch.empty((train.n,), dtype=torch.int64, device=dev)
        if nparts > 1:
            side = torch.cuda.Stream(device=dev, priority=SIDE_STREAM_PRIORITY)
            side_ws = torch.empty(int(be.lib.lvs_kmeans_accumulate_workspace_bytes(train.n, k)) + 256, dtype=torch.uint8, device=dev)
            cuts = range_cuts(train.n, fracs)
        for it in range(niter):
            if timed:
                ev = torch.cuda.Event(enable_timing=True)
                ev.record(torch.cuda.current_stream(dev))
                marks.append(ev)
            if nparts > 1:
                # ranges of rows: search range i, then its sums on the side stream while range i + 1 is searched
                main = torch.cuda.current_stream(dev)
                sums = torch.zeros((k, d), dtype=torch.float32, device=dev)
                counts = torch.zeros((k,), dtype=torch.float32, device=dev)
                held, kparts, last = [], [], None

                def settle(sub, handle, searched_ev):
                    # everything of a range that follows its one-pass search - the host's read of the certificate's two counts,
                    # the exact dot products of the pairs, the exact search of the open rows, the in-row-order sums - on the
                    # side stream, while the main stream already runs the next range's search
                    nonlocal last
                    with torch.cuda.stream(side):
                        side.wait_event(searched_ev)
                        kp = be.nearest_finish(handle, stats=stats)
                        be.kmeans_accumulate_keys_into(sub, kp, k, sums, counts, workspace=side_ws)
                        last = side.record_event()
                    kparts.append(kp)

                pending = None
                for i in range(nparts):
                    sub = be.slice_rows(train, cuts[i], cuts[i + 1])
                    if pipelined:
                        handle = be.nearest_begin(cpk, sub, _capi.METRIC_L2, exact_scores=False, corpus_stats=cstats)
                        ev = main.record_event()
                        held.append((sub, handle))  # alive until the side stream is done with them (the wait below)
                        if pending is not None:
                            settle(*pending)  # range i - 1 settles while range i's search (already queued) runs
                        pending = (sub, handle, ev)
                        continue
                    kp = be.nearest(cpk, sub, _capi.METRIC_L2, exact_scores=False, corpus_stats=cstats, stats=stats)
                    searched_ev = main.record_event()
                    with torch.cuda.stream(side):
                        side.wait_event(searched_ev)
                        be.kmeans_accumulate_keys_into(sub, kp, k, sums, counts, workspace=side_ws)
                        last = side.record_event()
                    held.append(kp)  # alive until the side stream is done with it (the wait below)
                    kparts.append(kp)
                if pending is not None:
                    settle(*pending)
                main.wait_event(last)
                if trace is not None:
                    trace.append({"centroids": centroids.clone(), "keys": torch.cat(kparts)})
                del held, kparts, pending
            elif not use_bounds and use_iter_op:
                # the whole iteration - assignment, sums, objective, all-reduce, division + split + repack - is ONE C-ABI call
                # (lvs_kmeans_iteration): the same launches in the same order as the branches below issue one by one
                if trace is not None:
                    c_before = centroids.clone()
                be.kmeans_iteration(train, x2, k, nt, centroids, cpk, cstats, iter_keys, obj_dev[it:it + 1], nsplit_dev[it:it + 1],
                                    all_reduce=(lambda t: _dist.all_reduce_sum_([t], process_group)) if dist is not None else None,
                                    stats=stats)
                if trace is not None:
                    trace.append({"centroids": c_before, "keys": iter_keys.clone().reshape(-1, 1)})
                continue
            elif not use_bounds:
                keys = be.nearest(cpk, train, _capi.METRIC_L2, exact_scores=False, corpus_stats=cstats, stats=stats)  # ids only ...
            else:
                act = None
                if keys is not None:  # bounds moved by the last update: which rows may have changed their centroid?
                    act = be.kmeans_bounds_step(b_assign, b_ub, b_lb, shift, top2)
                if keys is None or int(act.numel()) > train.n // 2:  # (almost) everything moved: search all rows in place
                    keys = be.nearest(cpk, train, _capi.METRIC_L2, exact_scores=False, corpus_stats=cstats, stats=stats,
                                      bounds=(b_assign, b_ub, b_lb, None))
                    searched = train.n
                elif int(act.numel()):
                    sub = be.gather(train, act)
                    keys[act] = be.nearest(cpk, sub, _capi.METRIC_L2, exact_scores=False, corpus_stats=cstats, stats=stats,
                                           bounds=(b_assign, b_ub, b_lb, act))
                    searched = int(act.numel())
                else:
                    searched = 0
                if stats is not None:
                    stats.setdefault("searched_rows", []).append(searched)
                c_old = centroids.clone()
            if nparts == 1:
                if trace is not None:
                    trace.append({"centroids": centroids.clone(), "keys": keys.clone()})
                sums, counts = be.kmeans_accumulate_keys(train, keys, k)
            # ... because the objective (faiss: sum of the assignment distances) follows from the sums the update needs
            # anyway:  sum_i |x_i - c_a(i)|^2 = sum_i |x_i|^2 - 2 sum_j c_j . S_j + sum_j n_j |c_j|^2   (float64, [k,d])
            be.kmeans_objective(centroids, sums, counts, x2, obj_dev[it:it + 1])
            if dist is not None:
                _dist.all_reduce_sum_([sums, counts, obj_dev[it:it + 1]], process_group)
            # centroid division + faiss split_clusters (same RNG stream on every rank) + repack, nothing read back
            cpk, cstats = be.kmeans_finish(sums, counts, centroids, nt, cmode, nsplit_dev[it:it + 1], exp=pexp)
            if use_bounds:
                shift, top2 = be.kmeans_centroid_shift(c_old, centroids)
        if timed and marks:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record(torch.cuda.current_stream(dev))
            ev.synchronize()
            marks.append(ev)
            stats["iteration_ms"] = [float(a.elapsed_time(b)) for a, b in zip(marks[:-1], marks[1:])]
        obj[:] = (obj_dev[:niter].cpu().numpy() * 2.0 ** (-2 * pexp)).astype(np.float32)
        nsplit[:] = nsplit_dev[:niter].cpu().numpy()
    assign = np.zeros(0, np.int64)
    if final_assign:
        cpk, cstats = be.kmeans_pack_centroids(centroids, cmode, exp=pexp)
        if dist is None:
            keys = be.nearest(cpk, packed, _capi.METRIC_L2, exact_scores=False, corpus_stats=cstats)  # ids only: no rescoring pass
            _, I = be.keys_to_result(keys, _capi.METRIC_L2)
            assign = np.asarray(I.reshape(-1).cpu().numpy(), dtype=np.int64)  # already int64: no copy
        else:
            import torch

            if sharded_rows:
                mine, pos = packed, local_pos
            else:  # replicated rows: contiguous row slices
                per = -(-n // world)
                lo, hi = min(n, rank * per), min(n, (rank + 1) * per)
                mine, pos = be.slice_rows(packed, lo, hi), np.arange(lo, hi, dtype=np.int64)
            keys = be.nearest(cpk, mine, _capi.METRIC_L2, exact_scores=False, corpus_stats=cstats)
            _, I = be.keys_to_result(keys, _capi.METRIC_L2)
            # ranks may hold different numbers of rows: exchange (position, cluster id) pairs padded to the largest share
            cnt = torch.tensor([mine.n], dtype=torch.int64, device=I.device)
            cmax = int(_dist.all_gather_rows(cnt, process_group).max().item())
            pair = torch.full((2, cmax), -1, dtype=torch.int64, device=I.device)
            pair[0, :mine.n] = be.to_device(pos)
            pair[1, :mine.n] = I.reshape(-1)
            allp = _dist.all_gather_rows(pair, process_group).cpu().numpy()  # [world, 2, cmax]
            assign = np.full(n, -1, np.int64)
            for r in range(world):
                ok = allp[r, 0] >= 0
                assign[allp[r, 0][ok]] = allp[r, 1][ok]
    cent = np.asarray(centroids.cpu().numpy(), np.float32)
    if pexp:
        cent = (cent * np.float32(2.0 ** -pexp)).astype(np.float32)  # exact
    return KMeansResult(centroids=cent, assign=assign, obj=obj, nsplit=nsplit, train_ids=train_ids)


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
        if hasattr(vs, "kmeans"):  # HipVS: the rows are already in HBM (possibly sharded) - no host copy of the matrix
            res = vs.kmeans(None, ncentroids, niter=niter, ids=ids, return_result=True)
        else:
            res = kmeans(vs.get_vectors_from_index(col_index_dir, ids), ncentroids, niter=niter,
                         backend=getattr(vs, "backend", None))
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
