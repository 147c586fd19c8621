"""``HipVS`` - MI355X-native drop-in for the reference's default vector store.

Mirrors ``lotus.vector_store.faiss_vs.FaissVS`` (``lotus/vector_store/faiss_vs.py:13-77``): same four plugin
methods (``lotus/vector_store/vs.py:17,24,31,54``), same on-disk ``{index_dir}/vecs`` pickle (+ a faiss-format
``{index_dir}/index``), same result conventions - ``RMOutput(distances float32 [Q,K], indices int64 [Q,K])`` best
first, missing slots ``-1`` / ``-FLT_MAX`` (inner product) or ``+FLT_MAX`` (L2).  The arithmetic runs in
``liblotus_hip.so`` (tiled MFMA distance + fused top-k); there is no CPU path.

Differences that are deliberate supersets (SURVEY.md section 8(a) edge-case table):
  * ``K == 0`` returns empty ``[Q,0]`` arrays instead of faiss's ``AssertionError``;
  * a dimension mismatch raises ``ValueError``;
  * ``ids`` covering every row in order skips the gather; a strict subset is gathered on the GPU instead of
    rebuilding an index from a re-read pickle (``faiss_vs.py:57-64``);
  * several indexes stay resident (keyed by ``index_dir``) so ``sem_sim_join`` flipping left/right
    (``sem_sim_join.py:111-128``) does not reload from disk;
  * with ``torch.distributed`` initialised and ``shard=True`` the corpus is row-sharded over the ranks and the
    per-shard top-k lists are merged after one all-gather (RCCL over xGMI on GPUs).
"""
from __future__ import annotations

import os
import pickle
from collections import OrderedDict
from dataclasses import dataclass
from typing import Any

import numpy as np

from . import _capi, store
from .compat import VS, RMOutput

METRIC_INNER_PRODUCT = _capi.METRIC_IP  # == faiss.METRIC_INNER_PRODUCT (0)
METRIC_L2 = _capi.METRIC_L2  # == faiss.METRIC_L2 (1)

FLT_MAX = np.float32(3.4028234663852886e38)
_RANK_BLOCK_SCORES = 1 << 28  # K = N ranking: scores (float32) materialised at a time


@dataclass
class _Resident:
    """One loaded index: its device image + where ``get_vectors_from_index`` gets the stored rows from."""

    vecs: Any  # [n,d] rows in their stored dtype: the caller's ndarray, a read-only memmap, or None (opened on demand)
    packed: Any  # backend PackedRows of this rank's shard
    n: int  # rows in the whole index
    d: int
    lo: int  # first global row of this rank's shard
    hi: int
    sig: Any = None  # store.signature() of the directory when it was loaded (None: never persisted)


def _serialised(fn):
    import functools

    @functools.wraps(fn)
    def run(self, *a, **k):
        with self._lock:
            shared = getattr(self._backend, "_call_lock", None)  # stores sharing one backend share its workspace too
            if shared is None:
                return fn(self, *a, **k)
            with shared:
                return fn(self, *a, **k)

    return run


class HipVS(VS):
    """Exact (brute-force) vector store on MI355X.

    Args:
        metric: ``METRIC_INNER_PRODUCT`` (default, as ``FaissVS``) or ``METRIC_L2`` (squared L2, ascending).
        storage: ``"auto"`` - fp16 embeddings are stored as fp16, fp32/fp64 embeddings as an fp16 hi|lo pair
            (fp32-accurate scores); ``"fp16"`` - round everything to fp16 (half the HBM, 3x the speed,
            ~1e-4 score error on fp32 inputs); ``"fp32"`` - always the hi|lo pair.
        device: torch device string; default current CUDA device.
        shard: how a join is split over the ``torch.distributed`` ranks.  ``True`` / ``"rows"`` - row-shard the corpus,
            queries replicated, one all-gather of the per-shard candidate keys + merge (BASELINE's configuration; scales the
            corpus beyond one GPU's HBM).  ``"queries"`` - every rank keeps the WHOLE corpus and searches its slice of the
            queries; the finished lists are all-gathered, nothing is merged.  ``(gq, gc)`` - the 2-D split: ``gq`` query
            groups x ``gc`` corpus shards (``gq * gc`` = ranks; rank r is in query group r // gc and holds corpus shard
            r % gc): every GPU searches Q / gq queries against N / gc rows, the lists are merged inside a corpus group and
            concatenated across the query groups.  ``"auto"`` - ``lotus_amd.plan.pick_split`` ranks the splits by the
            per-GPU shapes' measured rates (at configs[2] on 8 GPUs all four splits run within a point of 37 % of the
            MFMA roof, so the tie goes to the row split: least HBM per GPU).
        normalize: L2-normalise every row and every query on the device while packing (``lvs_pack_rows(normalize=1)``):
            with the inner-product metric this IS cosine similarity whatever the embedder returns.  ``FaissVS`` has no
            such switch - it relies on the RM normalising (``sentence_transformers_rm.py:30,71``) - so the default is off.
        max_resident: how many indexes stay on the GPU.
        backend: injected device backend (tests); default ``HipBackend``.
    """

    def __init__(self, metric: int = METRIC_INNER_PRODUCT, storage: str = "auto", device: str | None = None,
                 shard: bool | str = False, max_resident: int = 4, backend=None, process_group=None,
                 normalize: bool = False, abi_exchange: bool = False) -> None:
        super().__init__()
        # True: a row-sharded search runs its two exchanges from inside the C ABI (lvs_search_sharded, the process group's
        # all-gather handed over as a callback) instead of from this file - same kernels, same order, same result
        self.abi_exchange = bool(abi_exchange)
        if metric not in (METRIC_INNER_PRODUCT, METRIC_L2):
            raise ValueError("metric must be METRIC_INNER_PRODUCT or METRIC_L2")
        if storage not in ("auto", "fp16", "fp32"):
            raise ValueError("storage must be 'auto', 'fp16' or 'fp32'")
        self.metric = metric
        self.normalize = bool(normalize)
        self.storage = storage
        self.index_dir: str | None = None
        self._device = device
        self._backend = backend
        self._resident: "OrderedDict[str, _Resident]" = OrderedDict()
        self._max_resident = max(1, int(max_resident))
        is_pair = (isinstance(shard, (tuple, list)) and len(shard) == 2 and all(isinstance(v, int) and v >= 1 for v in shard))
        if not is_pair and shard not in (False, True, "rows", "queries", "auto"):
            raise ValueError("shard must be False, True, 'rows', 'queries', 'auto' or a (query groups, corpus shards) pair")
        self._split = tuple(shard) if is_pair else shard
        self._shard = shard in (True, "rows")      # corpus rows split across the ranks
        self._shard_queries = shard == "queries"   # corpus replicated, queries split
        self._pg = process_group
        self._lay = None                           # resolved layout, see _layout()
        import threading

        # one call at a time per store: the reference's VS is a process-global with mutable state too, and its only concurrent
        # caller (sem_topk's group-by thread pool, sem_topk.py:770-773 -> sem_index / sem_search) must not interleave the
        # launches of two searches that share one device workspace
        self._lock = threading.RLock()

    # ------------------------------------------------------------------------------------------------ helpers
    @property
    def backend(self):
        if self._backend is None:
            from .backend import HipBackend

            self._backend = HipBackend(self._device)
        return self._backend

    def _layout(self, sizes=None):
        """-> (qg, gq, cs, gc, pg_query, pg_corpus): this rank's query group / corpus shard under the configured split and
        the process groups its two exchange steps run in (None: nothing to exchange in that direction).  Resolved once,
        at first use: a 2-D split creates its sub-groups here (a collective call - every rank reaches it together, like
        every other step of a sharded operator).  ``sizes`` = (rows, d, bytes per stored value) of the index being
        installed: what ``shard="auto"`` plans with (the split is fixed by the FIRST index this store installs; without
        sizes - a search before any index - "auto" falls back to the row split, the only one that fits any corpus)."""
        if self._lay is not None:
            return self._lay
        if self._split is False:
            return (0, 1, 0, 1, None, None)
        import torch.distributed as dist

        if not (dist.is_available() and dist.is_initialized()):
            return (0, 1, 0, 1, None, None)
        rank, world = dist.get_rank(self._pg), dist.get_world_size(self._pg)
        split = self._split
        if split == "auto":
            from .plan import pick_split

            if sizes is None:
                split = (1, world)
            else:
                split = pick_split(world, nb=int(sizes[0]), d=int(sizes[1]), bytes_per_value=int(sizes[2]))
        if split in (True, "rows"):
            lay = (0, 1, rank, world, None, self._pg)
        elif split == "queries":
            lay = (rank, world, 0, 1, self._pg, None)
        else:
            gq, gc = int(split[0]), int(split[1])
            if gq * gc != world:
                raise ValueError(f"shard={split}: {gq} query groups x {gc} corpus shards need {gq * gc} ranks, the group has {world}")
            if gq == 1:
                lay = (0, 1, rank, world, None, self._pg)
            elif gc == 1:
                lay = (rank, world, 0, 1, self._pg, None)
            else:
                members = dist.get_process_group_ranks(self._pg) if self._pg is not None else list(range(world))
                qg, cs = rank // gc, rank % gc
                pg_corpus = pg_query = None
                for g in range(gq):  # every rank creates every group, in the same order
                    grp = dist.new_group([members[g * gc + c] for c in range(gc)])
                    if g == qg:
                        pg_corpus = grp
                for c in range(gc):
                    grp = dist.new_group([members[g * gc + c] for g in range(gq)])
                    if c == cs:
                        pg_query = grp
                lay = (qg, gq, cs, gc, pg_query, pg_corpus)
        self._lay = lay
        return lay

    def _dist(self):
        """(rank, world) of the corpus sharding - which contiguous block of rows this rank holds; (0, 1) when every rank
        holds all rows."""
        lay = self._layout()
        return lay[2], lay[3]

    def _pg_corpus(self):
        return self._layout()[5]

    def _group(self):
        """(rank, world) of the process group this store works in under EITHER split (rows or queries) - what decides who
        writes an index directory and which collectives every rank must enter; (0, 1) when not distributed."""
        if self._split is False:
            return 0, 1
        from . import _dist

        _, rank, world = _dist.context(True, self._pg)
        return rank, world

    def _pack_mode(self, dtype) -> int:
        if self.storage == "fp16":
            return _capi.PACK_F16
        if self.storage == "fp32":
            return _capi.PACK_SPLIT
        return _capi.PACK_F16 if np.dtype(dtype) == np.float16 else _capi.PACK_SPLIT

    @staticmethod
    def _is_device_tensor(x) -> bool:
        return hasattr(x, "is_cuda") and hasattr(x, "data_ptr") and bool(x.is_cuda)

    @staticmethod
    def _as_matrix(x, what: str):
        if HipVS._is_device_tensor(x):  # embeddings that never left the GPU
            if x.dim() == 1:
                x = x[None, :]
            if x.dim() != 2:
                raise ValueError(f"{what} must be a 2-D tensor, got shape {tuple(x.shape)}")
            return x
        x = np.asarray(x)
        if x.ndim == 1:
            x = x[None, :]
        if x.ndim != 2:
            raise ValueError(f"{what} must be a 2-D array, got shape {x.shape}")
        if x.dtype not in (np.float16, np.float32, np.float64):
            x = x.astype(np.float32)
        return x

    def _install(self, index_dir: str, vecs, stored=None, sig=None) -> _Resident:
        """Build the device image of this rank's shard from ``vecs`` ([n,d] ndarray / memmap / CUDA tensor; only rows
        [lo, hi) are touched).  ``stored``: what ``get_vectors_from_index`` serves (None = open on demand)."""
        n, d = int(vecs.shape[0]), int(vecs.shape[1])
        is_dev = self._is_device_tensor(vecs)
        dtype = np.float16 if (is_dev and str(vecs.dtype) == "torch.float16") else (np.float32 if is_dev else vecs.dtype)
        mode = self._pack_mode(dtype)
        self._layout(sizes=(n, d, 2 if mode == _capi.PACK_F16 else 4))  # "auto" plans with THIS index's real size
        rank, world = self._dist()
        per = -(-n // world) if n else 0
        lo, hi = min(n, rank * per), min(n, (rank + 1) * per)
        if world > 1 and mode == _capi.PACK_SPLIT and not self.normalize:
            packed = self._pack_shard_agreed(vecs, lo, hi, mode, is_dev)
        else:  # fp32-accurate rows are stored as x * 2^e with e chosen from the data (backend.pack); fp16 rows as given
            packed = self.backend.pack(vecs[lo:hi], mode, normalize=self.normalize, exp="auto", check=True)
        ent = _Resident(vecs=stored, packed=packed, n=n, d=d, lo=lo, hi=hi, sig=sig)
        self._resident[index_dir] = ent
        self._resident.move_to_end(index_dir)
        while len(self._resident) > self._max_resident:
            self._resident.popitem(last=False)
        return ent

    _EXP_HEAD_ROWS = 65536  # rows per shard the agreed exponent is sampled from

    def _pack_shard_agreed(self, vecs, lo: int, hi: int, mode: int, is_dev: bool):
        """Pack rows [lo, hi) of a row-sharded fp32-accurate index with ONE power-of-two scale for every shard (per-shard
        lists are merged by score).  The exponent is agreed from each rank's first rows (one tiny all-gather) and has 500x
        headroom; should some later row of some shard still leave fp16's range under it, every rank packs with the
        exponent of its own true maximum and the smallest of them is adopted.  Every rank of the corpus group enters every
        exchange, also one whose shard is empty - the decisions are taken from gathered values only."""
        import torch
        from . import _dist

        be, pg = self.backend, self._pg_corpus()

        def gathered(values):
            t = torch.tensor([float(v) for v in values], dtype=torch.float64)
            return _dist.all_gather_rows(t, pg).numpy()

        head = vecs[lo:min(hi, lo + self._EXP_HEAD_ROWS)]
        head = head.float().cpu().numpy() if is_dev else np.asarray(head, dtype=np.float32)
        amax = float(max(head.max(initial=0.0), -head.min(initial=0.0)))
        exp = be.exp_for(float(gathered([amax if np.isfinite(amax) else 0.0]).max()))
        packed = be.pack(vecs[lo:hi], mode, exp=exp, check="lazy")
        flags = gathered([int(packed.flags.item())]).astype(np.int64).reshape(-1)
        f = int(np.bitwise_or.reduce(flags))
        if f & _capi.PACK_FLAG_RANGE and not f & _capi.PACK_FLAG_NONFINITE:
            packed = be.pack(vecs[lo:hi], mode, exp="auto", check="lazy")
            mine = gathered([packed.exp, hi - lo, int(packed.flags.item())]).reshape(-1, 3)
            f = int(np.bitwise_or.reduce(mine[:, 2].astype(np.int64)))
            held = mine[mine[:, 1] > 0]
            exp = int(held[:, 0].min()) if len(held) else 0
            if not f and len(held) and (held[:, 0] != exp).any():  # decided from gathered values: same on every rank
                if packed.exp != exp:
                    packed = be.pack(vecs[lo:hi], mode, exp=exp, check="lazy")
                f = int(np.bitwise_or.reduce(gathered([int(packed.flags.item())]).astype(np.int64).reshape(-1)))
        be.raise_for_flags(f)
        packed.flags = None
        if hi == lo:
            packed.exp = int(exp)  # a rank without rows packs its queries with, and decodes merged keys by, the agreed exponent
        return packed

    def _current(self) -> _Resident:
        if self.index_dir is None or self.index_dir not in self._resident:
            raise ValueError("Index not loaded")  # faiss_vs.py:54-55
        return self._resident[self.index_dir]

    # ---------------------------------------------------------------------------------------- plugin methods
    @_serialised
    def index(self, docs, embeddings, index_dir: str, **kwargs: dict[str, Any]) -> None:
        """Build the index from ``embeddings`` and persist it (``faiss_vs.py:22-30``).  ``docs`` is unused, as in
        ``FaissVS``.  ``embeddings`` may also be a CUDA tensor straight from an encoder (no host round trip for the
        device image).  Rank 0 writes the reference's two files plus the mappable row store (``lotus_amd/store.py``);
        ``persist=False`` skips the disk entirely, ``raw=False`` writes the reference's files only."""
        emb = self._as_matrix(embeddings, "embeddings")
        rank, world = self._group()  # under the query split every rank holds the whole corpus, but only ONE may write it
        persist = bool(kwargs.get("persist", True))
        is_dev = self._is_device_tensor(emb)
        if persist:
            if rank == 0:
                host = emb.cpu().numpy() if is_dev else emb
                store.write_dir(index_dir, embeddings, host, self.metric, raw=bool(kwargs.get("raw", True)))
            if world > 1:  # the other ranks may open the directory right after this call
                from . import _dist

                _dist.barrier(self._pg)
        # the signature is taken after the barrier: every rank records the finished directory
        self._install(index_dir, emb, stored=None if is_dev else emb,
                      sig=store.signature(index_dir) if persist else None)
        self.index_dir = index_dir

    @_serialised
    def load_index(self, index_dir: str) -> None:
        """Make ``index_dir`` the current index (``faiss_vs.py:32-36``).  Served from HBM when the directory is already
        resident AND unchanged on disk since it was loaded; otherwise this rank's rows are read through a memory map
        (nothing is unpickled, ``lotus_amd/store.py``) and packed."""
        ent = self._resident.get(index_dir)
        if ent is not None and (ent.sig is None or ent.sig == store.signature(index_dir)):
            self._resident.move_to_end(index_dir)
            self.index_dir = index_dir
            return
        sig = store.signature(index_dir)
        rows, _ = store.open_device_rows(index_dir)
        rows = self._as_matrix(rows, "stored vectors")
        self._install(index_dir, rows, stored=None, sig=sig)
        self.index_dir = index_dir

    @_serialised
    def get_vectors_from_index(self, index_dir: str, ids) -> np.ndarray:
        """``vecs[ids]`` in the stored dtype (``faiss_vs.py:38-41``); ``ids`` may be a list or a pandas Index.  Rows
        come from the caller's array (same process), from the row store's memory map (only the touched pages are read)
        or - for an index that was never persisted - from the device image."""
        ent = self._resident.get(index_dir)
        sel = ids if isinstance(ids, slice) else np.asarray(ids, dtype=np.int64)
        if ent is not None and ent.vecs is None and ent.sig is None:  # persist=False / device tensors: HBM is the store
            if ent.lo != 0 or ent.hi != ent.n:
                raise NotImplementedError("vectors of a sharded index that was never persisted")  # sem_sim_join.py:114-117
            be = self.backend
            if isinstance(sel, slice):
                sel = np.arange(ent.n, dtype=np.int64)[sel]
            out = be.unpack(ent.packed, be.to_device(sel)).cpu().numpy()
            return out.astype(np.float16) if ent.packed.mode == _capi.PACK_F16 else out
        if ent is not None and ent.vecs is not None and (ent.sig is None or ent.sig == store.signature(index_dir)):
            vecs = ent.vecs  # the caller's array / an open map - unless the directory was rewritten since (load_index's rule)
        else:
            vecs, _ = store.open_stored_rows(index_dir)
            if ent is not None and ent.sig == store.signature(index_dir):
                ent.vecs = vecs
        return np.asarray(vecs[sel])

    @_serialised
    def __call__(self, query_vectors, K: int, ids: list[int] | None = None, **kwargs: dict[str, Any]) -> RMOutput:
        """Top-``K`` rows for every query vector (``faiss_vs.py:43-77``)."""
        ent = self._current()
        q = self._as_matrix(query_vectors, "query_vectors")
        if q.shape[1] != ent.d:
            raise ValueError(f"query dimension {q.shape[1]} does not match index dimension {ent.d}")
        K = int(K)
        if K < 0:
            raise ValueError("K must be >= 0")
        nq = int(q.shape[0])
        return_device = bool(kwargs.get("return_device", False))
        pad_d = -FLT_MAX if self.metric == METRIC_INNER_PRODUCT else FLT_MAX
        if K == 0 or nq == 0:
            return RMOutput(distances=np.full((nq, K), pad_d, np.float32), indices=np.full((nq, K), -1, np.int64))

        be = self.backend
        sub = None
        if ids is not None:
            sub = np.asarray(ids, dtype=np.int64).reshape(-1)
            if sub.size and (sub.min() < 0 or sub.max() >= ent.n):
                raise IndexError("ids out of range for the loaded index")
            if sub.size == ent.n and (ent.n == 0 or (sub[0] == 0 and sub[-1] == ent.n - 1
                                                     and np.array_equal(sub, np.arange(ent.n)))):
                sub = None  # every row, in order: same as an unfiltered search (sem_sim_join.py:132-134)
        n_eff = ent.n if sub is None else int(sub.size)
        k_eff = min(K, n_eff)
        if k_eff == 0:
            return RMOutput(distances=np.full((nq, K), pad_d, np.float32), indices=np.full((nq, K), -1, np.int64))
        rank_all = k_eff > _capi.MAX_K  # K = N callers (sem_dedup.py:45, sem_filter.py:491-497): full score rows + sort
        qrank, qworld, rank, world, pg_query, pg_corpus = self._layout()
        q_all = nq
        if qworld > 1:  # this rank's contiguous slice of the queries (possibly empty)
            per = -(-nq // qworld)
            q = q[min(nq, qrank * per):min(nq, (qrank + 1) * per)]
            nq = int(q.shape[0])

        # queries share the index's power-of-two scale (required for L2; for inner products it keeps one exponent per
        # index); they are validated while they are packed, the flag word comes back together with the results
        qexp = kwargs.get("_query_exp", ent.packed.exp)
        if (sub is None and not rank_all and world == 1 and qworld == 1 and not return_device and k_eff == K
                and not self._is_device_tensor(q) and qexp != "auto" and hasattr(be, "search_host_pipelined")
                and nq >= be.CALL_PIPELINE_MIN_QUERIES and not self._fp32_path(ent, q)):
            # the plain big call (sem_sim_join.py:132-134 -> faiss_vs.py:75): transfers overlapped with the search
            Dh, Ih, f = be.search_host_pipelined(ent.packed, q, k_eff, self.metric, id_offset=ent.lo,
                                                 normalize=self.normalize, exp=int(qexp))
            redo = self._check_queries(f, query_vectors, K, ids, kwargs)
            return redo if redo is not None else RMOutput(distances=Dh, indices=Ih)
        if qexp == "auto" and qworld > 1:
            # the finished lists of all query groups are decoded with ONE score exponent: agree it from the largest magnitude
            import torch
            from . import _dist

            amax = be.absmax(q) if (nq and self._is_device_tensor(q)) else float(np.abs(q).max(initial=0.0)) if nq else 0.0
            t = torch.tensor([amax if np.isfinite(amax) else 0.0], dtype=torch.float64)
            qexp = be.exp_for(float(_dist.all_gather_rows(t, pg_query).max())) if ent.packed.mode == _capi.PACK_SPLIT else 0
        queries = be.pack(q, ent.packed.mode, normalize=self.normalize, exp=qexp, check="lazy")
        score_exp = be.score_exp_of(ent.packed, queries)
        flags = getattr(queries, "flags", None)
        if qworld > 1 and flags is not None:
            # every query group validates its own slice; the verdict (raise / search again with another exponent - both
            # collective decisions) must be the same everywhere: OR of the flag words, one tiny all-gather
            from . import _dist

            g = _dist.all_gather_rows(flags, pg_query)
            flags = ((g & 1).amax(0) | (g & 2).amax(0)).to(flags.dtype)
        id_map = None
        if rank_all:
            # score rows of this rank's shard, exchanged so that every rank ranks the complete rows (column-sharded
            # score matrix, one all-gather); the device sort goes through the queries in chunks of < 2^32 scores
            # ... and in blocks of at most 2^28 scores (1 GB of float32), so that a K = N call never holds the whole Q x N
            # matrix next to its Q x N keys (sem_dedup's reference path asks for N x N)
            n_cols = ent.n if sub is None else int(sub.size)
            qstep = max(1, min(nq, _RANK_BLOCK_SCORES // max(1, n_cols)))
            parts, order = [], None
            for q0 in range(0, nq, qstep):
                sc, order = self._score_rows(ent, be.slice_rows(queries, q0, min(nq, q0 + qstep)), sub, world)
                parts.append(be.rank_scores(sc)[:, :k_eff].contiguous())
                del sc
            if len(parts) == 1:
                keys = parts[0]
            else:
                import torch

                keys = torch.cat(parts)
            score_exp = 0  # score rows come back in the caller's units already
            if order is not None:
                id_map = be.to_device(order)
            world = 1  # already complete on every rank: nothing left to merge
        elif sub is None and world > 1 and self.abi_exchange and hasattr(be, "search_sharded") and k_eff <= 56:
            # the same row-sharded search with its two exchanges issued from INSIDE the C ABI (lvs_search_sharded): the transport
            # is this process group's all-gather, handed over as a callback
            from . import _dist

            per = -(-ent.n // world) if ent.n else 0
            tiles = be.seed_tiles(queries.n, per, k_eff, ent.packed.mode, queries.mode) if hasattr(be, "seed_tiles") else 0
            keys = be.search_sharded(ent.packed, queries, k_eff, self.metric, ent.lo, world, max(0, tiles),
                                     lambda t: _dist.all_gather_rows(t, self._pg_corpus()))
            world = 1  # merged already
        elif sub is None:
            keys = be.search_keys(ent.packed, queries, k_eff, self.metric, id_offset=ent.lo,
                                  seed_scores=self._pooled_seed_scores(ent, queries, k_eff, world))
        else:
            # positions (in `ids`) of the subset rows that live in this rank's shard
            pos = np.flatnonzero((sub >= ent.lo) & (sub < ent.hi))
            local = be.to_device(sub[pos] - ent.lo)
            gathered = be.gather(ent.packed, local)
            row_ids = be.to_device(pos.astype(np.uint32).view(np.int32))
            keys = be.search_keys(gathered, queries, k_eff, self.metric, id_offset=0, row_ids=row_ids)
            id_map = be.to_device(sub)
        if world > 1:
            keys = self._allgather_merge(keys, world)
        if qworld > 1:  # finished lists of every query group's slice, side by side: one all-gather, no merge
            import torch
            from . import _dist

            per = -(-q_all // qworld)
            pad = torch.zeros((per, k_eff), dtype=keys.dtype, device=keys.device)
            pad[:nq] = keys
            keys = _dist.all_gather_rows(pad, pg_query).reshape(qworld * per, k_eff)[:q_all].contiguous()
            nq = q_all
        Dd, Id = be.keys_to_result(keys, self.metric, id_map, score_exp=score_exp)
        if return_device and k_eff == K:  # results stay in HBM (torch tensors) for a GPU-side consumer
            if flags is not None:
                redo = self._check_queries(int(flags.item()), query_vectors, K, ids, kwargs)
                if redo is not None:
                    return redo
            return RMOutput(distances=Dd, indices=Id)
        if hasattr(be, "to_host"):  # all copies in flight together, one synchronisation, pinned-backed result arrays
            Dh, Ih, *fh = be.to_host(Dd, Id, *([flags] if flags is not None else []))
        else:
            Dh, Ih, *fh = [t.cpu().numpy() for t in ((Dd, Id) + ((flags,) if flags is not None else ()))]
        if fh:
            redo = self._check_queries(int(fh[0][0]), query_vectors, K, ids, kwargs)
            if redo is not None:
                return redo
        if k_eff == K:
            return RMOutput(distances=Dh, indices=Ih)
        D = np.full((nq, K), pad_d, np.float32)  # fewer than K rows exist: faiss pads with -1 / -+FLT_MAX (Appendix A.2)
        I = np.full((nq, K), -1, np.int64)
        D[:, :k_eff] = Dh
        I[:, :k_eff] = Ih
        return RMOutput(distances=D, indices=I)

    @staticmethod
    def _fp32_path(ent, q) -> bool:
        """fp32-accurate (hi|lo) operands take the certified one-pass search, which reads a count back mid-call: the staged
        pipeline is for the plain fp16 search."""
        return ent.packed.mode != _capi.PACK_F16

    def _check_queries(self, f: int, query_vectors, K, ids, kwargs):
        """Validation flags of the packed queries (``lvs_pack_rows_checked``).  inf / NaN raise.  Magnitudes that leave
        fp16's range under the INDEX's scale are searched again with an exponent of their own when the metric allows it
        (inner products; squared L2 needs one scale on both sides): the offending queries on their own, so that the rest
        of the batch keeps the index's exponent and its full precision - returns that result, else None.  The split is
        taken from the complete (replicated) query matrix, so every rank of a sharded store makes the same calls."""
        if not f:
            return None
        ent = self._current()
        if not (f & _capi.PACK_FLAG_RANGE and not f & _capi.PACK_FLAG_NONFINITE and self.metric == METRIC_INNER_PRODUCT
                and "_query_exp" not in kwargs and ent.packed.mode == _capi.PACK_SPLIT):
            self.backend.raise_for_flags(f, "query vectors")
            return None
        q = self._as_matrix(query_vectors, "query_vectors")
        on_device = self._is_device_tensor(q)
        big = self._out_of_range_rows(q, ent)
        kw = dict(kwargs)
        kw["_query_exp"] = "auto"
        if big.all() or not big.any():
            return self.__call__(q, K, ids, **kw)
        sel_big, sel_rest = np.flatnonzero(big), np.flatnonzero(~big)
        if on_device:
            import torch

            pick = lambda sel: q[torch.from_numpy(sel).to(q.device)]
        else:
            pick = lambda sel: q[sel]
        rest = self.__call__(pick(sel_rest), K, ids, **kwargs)
        own = self.__call__(pick(sel_big), K, ids, **kw)
        if self._is_device_tensor(rest.distances) or hasattr(rest.distances, "index_copy_"):  # return_device=True
            import torch

            D = torch.empty((len(big), K), dtype=rest.distances.dtype, device=rest.distances.device)
            I = torch.empty((len(big), K), dtype=rest.indices.dtype, device=rest.indices.device)
            for sel, part in ((sel_rest, rest), (sel_big, own)):
                at = torch.from_numpy(sel).to(D.device)
                D[at], I[at] = part.distances, part.indices
            return RMOutput(distances=D, indices=I)
        D, I = np.empty((len(big), K), np.float32), np.empty((len(big), K), np.int64)
        D[sel_rest], I[sel_rest] = rest.distances, rest.indices
        D[sel_big], I[sel_big] = own.distances, own.indices
        return RMOutput(distances=D, indices=I)

    def _out_of_range_rows(self, q, ent) -> np.ndarray:
        """bool [nq]: queries with a component beyond fp16's range under the index's power-of-two scale."""
        rowmax = q.abs().amax(dim=1).float().cpu().numpy() if self._is_device_tensor(q) else np.abs(q).max(axis=1, initial=0.0)
        return np.asarray(rowmax, dtype=np.float64) * 2.0 ** ent.packed.exp > 65504.0

    @_serialised
    def scores(self, query_vectors, ids: list[int] | None = None, _query_exp=None):
        """Similarity of every query to every indexed row (or to rows ``ids``, in that order) as one float32 matrix
        [Q, N] - what the K = N callers actually want (``sem_filter.py:491-497`` takes ``vec_scores`` of ALL rows,
        ``sem_join.py:343-373`` clips them to [0, 1]) without ranking anything.  Inner product:
        the product; L2: minus the squared distance.  On a sharded index every rank computes the columns of its
        shard and one all-gather completes the rows."""
        ent = self._current()
        q = self._as_matrix(query_vectors, "query_vectors")
        if q.shape[1] != ent.d:
            raise ValueError(f"query dimension {q.shape[1]} does not match index dimension {ent.d}")
        be = self.backend
        sub = None
        if ids is not None:
            sub = np.asarray(ids, dtype=np.int64).reshape(-1)
            if sub.size and (sub.min() < 0 or sub.max() >= ent.n):
                raise IndexError("ids out of range for the loaded index")
            if sub.size == ent.n and np.array_equal(sub, np.arange(ent.n)):
                sub = None
        _, world = self._dist()
        queries = be.pack(q, ent.packed.mode, normalize=self.normalize,
                          exp=ent.packed.exp if _query_exp is None else _query_exp, check="lazy")
        f = int(queries.flags.item()) if getattr(queries, "flags", None) is not None else 0
        if (f & _capi.PACK_FLAG_RANGE and not f & _capi.PACK_FLAG_NONFINITE and self.metric == METRIC_INNER_PRODUCT
                and _query_exp is None and ent.packed.mode == _capi.PACK_SPLIT):
            # magnitudes outside fp16's range under the index's scale: inner products allow those queries an exponent of
            # their own (the rest of the batch keeps the index's, as in __call__)
            big = self._out_of_range_rows(q, ent)
            if big.all() or not big.any():
                return self.scores(q, ids, _query_exp="auto")
            if self._is_device_tensor(q):
                import torch

                pick = lambda sel: q[torch.from_numpy(np.flatnonzero(sel)).to(q.device)]
            else:
                pick = lambda sel: q[sel]
            rest, own = self.scores(pick(~big), ids), self.scores(pick(big), ids, _query_exp="auto")
            out = np.empty((len(big), rest.shape[1]), np.float32)
            out[~big], out[big] = rest, own
            return out
        if f:
            be.raise_for_flags(f, "query vectors")
        sc, order = self._score_rows(ent, queries, sub, world, want_ids=False)
        out = sc.cpu().numpy()
        if order is not None:  # columns arrived shard by shard: put them back into the order of `ids`
            res = np.empty_like(out)
            res[:, order] = out
            out = res
        return out

    def _score_rows(self, ent: _Resident, queries, sub, world: int, want_ids: bool = True):
        """Score rows [Q, n_eff] (device float32) of `queries` against all rows (``sub`` None) or the rows ``sub``.

        Returns ``(scores, order)``.  Unsharded: columns follow ``sub`` (``order`` = ``sub`` when ids are wanted, else
        None).  Sharded: every rank scores the requested rows that live in its shard, the blocks are all-gathered and
        laid side by side in rank order; ``order[j]`` then names column j - the global row id (``want_ids``) or its
        position in ``sub`` - and is None when the columns are simply rows 0..n-1."""
        be = self.backend
        if world == 1:
            corpus = ent.packed if sub is None else be.gather(ent.packed, be.to_device(sub))
            sc = be.scores(corpus, queries, self.metric)
            return sc, (sub if (sub is not None and want_ids) else None)
        from . import _dist

        per = -(-ent.n // world) if ent.n else 0
        bounds = [(min(ent.n, r * per), min(ent.n, (r + 1) * per)) for r in range(world)]
        if sub is None:
            pos = [np.arange(lo, hi, dtype=np.int64) for lo, hi in bounds]  # global rows per rank
            corpus = ent.packed
        else:
            pos = [np.flatnonzero((sub >= lo) & (sub < hi)) for lo, hi in bounds]  # positions in `sub` per rank
            mine = pos[bounds.index((ent.lo, ent.hi))]
            corpus = be.gather(ent.packed, be.to_device(sub[mine] - ent.lo))
        widths = [len(p_) for p_ in pos]
        wmax = max(widths) if widths else 0
        nq = queries.n
        import torch

        block = torch.full((nq, wmax), float("-inf"), dtype=torch.float32, device=queries.rows.device)
        if corpus.n:
            block[:, :corpus.n] = be.scores(corpus, queries, self.metric)
        parts = _dist.all_gather_rows(block, self._pg_corpus())  # [world, nq, wmax]
        sc = torch.cat([parts[r][:, :widths[r]] for r in range(world)], dim=1).contiguous()
        if sub is None:
            return sc, None
        cols = np.concatenate(pos) if pos else np.zeros(0, np.int64)
        return sc, (sub[cols] if want_ids else cols)

    def packed_rows(self, ids=None):
        """Device image (backend ``PackedRows``) of the current index restricted to positional ``ids`` (all rows
        when ``ids`` is None or covers them in order) - what the GPU-side operators (dedup, k-means) consume.
        Needs the whole index on this rank (``shard=False``)."""
        ent = self._current()
        if ent.lo != 0 or ent.hi != ent.n:
            raise ValueError("packed_rows needs an unsharded index")
        if ids is None:
            return ent.packed
        sub = np.asarray(ids, dtype=np.int64).reshape(-1)
        if sub.size == ent.n and np.array_equal(sub, np.arange(ent.n)):
            return ent.packed
        if sub.size and (sub.min() < 0 or sub.max() >= ent.n):
            raise IndexError("ids out of range for the loaded index")
        return self.backend.gather(ent.packed, self.backend.to_device(sub))

    @_serialised
    def kmeans(self, vec_set, ncentroids: int, niter: int = 20, ids=None, return_result: bool = False, **kw):
        """faiss-parity k-means of the current index's rows ``ids`` (``lotus/utils.py:61-65``) on the GPU(s) that already
        hold them; returns the cluster id of every row (all rows on every rank).  ``vec_set`` is not needed (the device
        image is used) and only kept for the shape of the reference call.  On a row-sharded index every rank trains on
        and assigns the rows of its own shard (one all-reduce per iteration, one all-gather of the ids)."""
        from .cluster import kmeans as _kmeans

        ent = self._current()
        sub = None
        if ids is not None:
            sub = np.asarray(ids, dtype=np.int64).reshape(-1)
            if sub.size and (sub.min() < 0 or sub.max() >= ent.n):
                raise IndexError("ids out of range for the loaded index")
            if sub.size == ent.n and np.array_equal(sub, np.arange(ent.n)):
                sub = None
        be = self.backend
        # the path is chosen from the group size - identical on every rank - never from this rank's share of the rows
        # (with ceil(n / world) >= n rank 0 holds everything while the others still enter the collectives)
        if self._dist()[1] == 1:  # the whole index lives on this rank
            packed = ent.packed if sub is None else be.gather(ent.packed, be.to_device(sub))
            res = _kmeans(None, ncentroids, niter=niter, backend=be, packed=packed, process_group=self._pg_corpus(), **kw)
        else:
            if sub is None:
                packed, local_pos, n_total = ent.packed, np.arange(ent.lo, ent.hi, dtype=np.int64), ent.n
            else:
                local_pos = np.flatnonzero((sub >= ent.lo) & (sub < ent.hi))  # positions in `ids`, ascending
                packed = be.gather(ent.packed, be.to_device(sub[local_pos] - ent.lo))
                n_total = int(sub.size)
            kw.pop("shard", None)
            res = _kmeans(None, ncentroids, niter=niter, backend=be, packed=packed, shard=True,
                          process_group=self._pg_corpus(), n_total=n_total, local_pos=local_pos, **kw)
        return res if return_result else res.assign

    # ------------------------------------------------------------------------------------------ multi-GPU
    def _pooled_seed_scores(self, ent: _Resident, queries, k: int, world: int):
        """Row-sharded join: every shard scores a small sample of its own rows and the blocks are all-gathered inside the
        corpus group, so that every shard starts from thresholds that know ALL shards' samples (an 8 x larger sample at no
        extra MFMA cost per GPU; ``lvs_flat_search_keys_seeded``).  None when the shape is not worth a collective; the
        decision depends on the call's arguments and the nominal shard size only, so every rank takes it alike."""
        be = self.backend
        if world <= 1 or not hasattr(be, "seed_tiles"):
            return None
        per = -(-ent.n // world) if ent.n else 0
        tiles = be.seed_tiles(queries.n, per, k, ent.packed.mode, queries.mode)
        if tiles <= 0:
            return None
        from . import _dist

        mine = be.seed_scores(ent.packed, queries, self.metric, tiles)                    # [tiles, nq]
        return _dist.all_gather_rows(mine, self._pg_corpus()).reshape(world * tiles, queries.n)

    def _allgather_merge(self, keys, world: int):
        """All-gather the per-shard candidate keys [Q,k] (8 B each; ONE RCCL all-gather over xGMI on a GPU node, staged
        through the host for any other process-group backend) and merge them on every rank (``lvs_merge_keys``)."""
        from . import _dist

        return self.backend.merge_keys(_dist.all_gather_rows(keys, self._pg_corpus()))
