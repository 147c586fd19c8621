"""Device backend: torch provides HBM buffers, streams and (for multi-GPU) RCCL; every compute step is a C-ABI call
into liblotus_hip.so.  No CPU fallback - constructing the backend without a GPU raises."""
from __future__ import annotations

import ctypes
from dataclasses import dataclass

import numpy as np

from . import _capi
from ._capi import LotusHipError


@dataclass
class PackedRows:
    """Device image of an embedding matrix: fp16 rows [n, ld] (+ |x|^2 per row)."""

    rows: "object"  # torch.Tensor float16 [n, ld]
    norms: "object"  # torch.Tensor float32 [n]: |row|^2 of the STORED (scaled) values
    n: int
    d: int
    mode: int  # _capi.PACK_F16 | _capi.PACK_SPLIT
    exp: int = 0  # the rows hold x * 2^exp (an exact power-of-two scale chosen at pack time; 0 = as given)
    flags: "object" = None  # device int32 [1] of LVS_PACK_FLAG_* when the pack was validated lazily (check="lazy")


class _DevBytes:
    """A raw device pointer as something ``torch.as_tensor`` understands (the all-gather callback of ``lvs_search_sharded``)."""

    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {"shape": (int(nbytes),), "typestr": "|u1", "data": (int(ptr), False), "version": 2}


def _ptr(t) -> int:
    return 0 if t is None else int(t.data_ptr())


class HipBackend:
    PACK_CHUNK_ROWS = 262144
    NEAREST3_WS_BUDGET = 2 << 30  # bytes of scratch one lvs_nearest3 call may ask for; more queries go through in chunks
    MAX_STREAM_WORKSPACES = 4     # scratch buffers kept at a time (one per stream that has made a call)

    def __init__(self, device=None):
        import torch

        self.torch = torch
        self.lib = _capi.load()
        if not torch.cuda.is_available():
            raise LotusHipError("lotus_amd needs a ROCm GPU (torch.cuda.is_available() is False); no CPU fallback")
        self.device = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
        self._ws = None
        # host -> device staging (see _h2d): a pinned buffer, a copy stream and a few threads that fill the buffer
        self._stage_buf = None   # pinned ring [H2D_SLOTS, H2D_SLICE_BYTES]; False: pinning failed, plain copies
        self._stage_done = None  # per ring slot: event of the last DMA that read it
        self._copy_stream = None
        self._pool = None
        import threading

        self._h2d_lock = threading.Lock()
        # one device workspace and one set of side streams per backend: stores that share a backend take this lock around
        # every plugin call (HipVS), so two threads never interleave the launches of two operations
        self._call_lock = threading.RLock()

    # ---- plumbing ----
    def _stream(self) -> int:
        return int(self.torch.cuda.current_stream(self.device).cuda_stream)

    def _c(self, name: str, *args) -> None:
        """One C-ABI call on this backend's device.  The library makes the stream's device current itself, but torch's
        default stream is the NULL handle, which names no device - so the device is also made current here."""
        with self.torch.cuda.device(self.device):
            _capi.check(getattr(self.lib, name)(*args), name)

    def _workspace(self, nbytes: int):
        """The scratch buffer of the CURRENT stream (one per stream: launches queued on a side stream - the k-means sums and
        certificates of one row range under the assignment search of the next - never share scratch memory with the main one)."""
        if self._ws is None:
            self._ws = {}
        key = self._stream()
        ws = self._ws.get(key)
        if ws is None or ws.numel() < nbytes:
            if ws is None and len(self._ws) >= self.MAX_STREAM_WORKSPACES:
                # side streams come and go (one per k-means call): keep the default stream's buffer, drop the others - a dropped
                # buffer goes back to the allocator's pool of the stream it was allocated on, so work still queued there is safe
                for k_old in [k_ for k_ in self._ws if k_ != 0]:
                    del self._ws[k_old]
            ws = self._ws[key] = self.torch.empty(max(nbytes, 1 << 20), dtype=self.torch.uint8, device=self.device)
        return ws

    def synchronize(self) -> None:
        self.torch.cuda.synchronize(self.device)

    def to_device(self, arr: np.ndarray):
        return self.torch.from_numpy(np.ascontiguousarray(arr)).to(self.device)

    # ---- host <-> device transfers at the boundary (VS.__call__ hands over host ndarrays, faiss_vs.py:43-77) ----
    H2D_SLICE_BYTES = 16 << 20
    H2D_SLOTS = 4       # pinned ring: H2D_SLOTS x H2D_SLICE_BYTES = 64 MB whatever the matrix (a pack chunk may be GBs)
    H2D_THREADS = 4

    def _h2d_ring(self):
        """The pinned staging ring, its copy stream and thread pool (created once; None when pinning fails - e.g. under a
        container's memlock limit - and the plain ``.to()`` path is used from then on)."""
        torch = self.torch
        if self._stage_buf is False:
            return None
        if self._stage_buf is None:
            try:
                self._stage_buf = torch.empty((self.H2D_SLOTS, self.H2D_SLICE_BYTES), dtype=torch.uint8, pin_memory=True)
            except Exception:  # cannot page-lock: fall back for good
                self._stage_buf = False
                return None
            from concurrent.futures import ThreadPoolExecutor

            self._pool = ThreadPoolExecutor(self.H2D_THREADS)
            self._copy_stream = torch.cuda.Stream(device=self.device)
            self._stage_done = [None] * self.H2D_SLOTS
        return self._stage_buf

    def _h2d(self, c: np.ndarray, out=None, after=None):
        """``out`` / ``after``: copy into this preallocated device tensor, the side stream waiting for the event ``after``
        (recorded when ``out`` was allocated) instead of for everything queued on the compute stream - a transfer that must
        overlap a search already in flight (``search_host_pipelined``).

        Device copy of a C-contiguous float16 / float32 host matrix.  A pageable ndarray reaches the GPU through a
        pinned staging buffer either way; doing that staging here, in 16 MB slices through a small ring - a few threads copy
        the next slices into free ring slots (numpy releases the GIL) while the DMA engine moves the finished ones on a side
        stream - makes the transfer run at the slower of (multi-threaded memcpy, PCIe) instead of a single-threaded memcpy
        followed by the DMA: 154 MB of queries in ~4 ms instead of ~15-25.  The compute stream waits on an event.  The ring
        is bounded (64 MB) and guarded by a lock: two threads sharing a backend take turns."""
        torch = self.torch
        nbytes = int(c.nbytes)
        ring = None
        if nbytes >= (4 << 20) and c.ndim == 2 and int(c.shape[1]) * c.itemsize <= self.H2D_SLICE_BYTES:
            with self._h2d_lock:
                ring = self._h2d_ring()
        if ring is None:
            import warnings

            with warnings.catch_warnings():  # a read-only memory map (store.py): the host view is only read by the copy
                warnings.simplefilter("ignore", UserWarning)
                t = torch.from_numpy(c)
            if out is not None:
                out.copy_(t)
                return out
            return t.to(self.device)
        tdt = torch.float16 if c.dtype == np.float16 else torch.float32
        rows, row_bytes = int(c.shape[0]), int(c.shape[1]) * c.itemsize
        per = max(1, self.H2D_SLICE_BYTES // row_bytes)
        bounds = [(a, min(rows, a + per)) for a in range(0, rows, per)]
        with self._h2d_lock:
            dev = out if out is not None else torch.empty(c.shape, dtype=tdt, device=self.device)
            cur = torch.cuda.current_stream(self.device)
            cs = self._copy_stream
            if out is not None and after is not None:
                cs.wait_event(after)
            else:
                cs.wait_stream(cur)  # `dev` was allocated on the compute stream
            slots = [ring[i][:per * row_bytes].view(tdt).reshape(per, int(c.shape[1])) for i in range(self.H2D_SLOTS)]
            slots_np = [t.numpy() for t in slots]
            pending, nxt = [], 0

            def submit():  # keep up to H2D_SLOTS host copies in flight; a slot is reused once its previous DMA has finished
                nonlocal nxt
                while nxt < len(bounds) and len(pending) < self.H2D_SLOTS:
                    slot = nxt % self.H2D_SLOTS
                    if self._stage_done[slot] is not None:
                        self._stage_done[slot].synchronize()
                    a, b = bounds[nxt]
                    pending.append((nxt, self._pool.submit(np.copyto, slots_np[slot][:b - a], c[a:b])))
                    nxt += 1

            last = None
            with torch.cuda.stream(cs):
                while nxt < len(bounds) or pending:
                    submit()
                    i, f = pending.pop(0)
                    f.result()
                    a, b = bounds[i]
                    dev[a:b].copy_(slots[i % self.H2D_SLOTS][:b - a], non_blocking=True)
                    last = self._stage_done[i % self.H2D_SLOTS] = cs.record_event()
            dev.record_stream(cs)
            if last is not None:
                cur.wait_event(last)
        return dev

    def to_host(self, *tensors):
        """Device tensors -> numpy arrays backed by pinned host memory (torch's caching host allocator: no page-locking
        per call), all copies in flight together, ONE stream synchronisation."""
        torch = self.torch
        outs = []
        for t in tensors:
            h = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
            h.copy_(t, non_blocking=True)
            outs.append(h)
        torch.cuda.current_stream(self.device).synchronize()
        return tuple(h.numpy() for h in outs)

    CALL_PIPELINE_MIN_QUERIES = 32768
    CALL_CHUNK = 32768  # queries per launch of a chunked search (LVS_RQ_CHUNK_DEFAULT, lotus_amd/csrc/lvs_tile.h)
    CALL_PIPELINE = (0.1, 0.9)  # shares of the queries per stage: a short first stage (its H2D is the exposed one; measured at
    # 100 k x 1 M, tools/tcall_probe.py: one stage 144.0 ms, (0.1, 0.9) 139.2, (0.2, 0.8) 141.5, (0.5, 0.5) 145.1, three or four
    # stages 141.8-150.1 against 135.9 ms device-resident - shorter launches run further below the long-stream rate)

    def search_host_pipelined(self, corpus: PackedRows, q: np.ndarray, k: int, metric: int, id_offset: int = 0,
                              normalize: bool = False, exp: int = 0):
        """``VS.__call__`` for a large host query matrix (faiss_vs.py:75): the queries go through in two stages so that the
        host -> device copy of the second overlaps the search of the first and the device -> host copy of the first stage's
        results overlaps the search of the second (results are per query: staging is exact).  Only the first stage's H2D
        and the last stage's D2H are exposed.  -> (D float32 [nq,k], I int64 [nq,k] pinned-backed numpy arrays, flag word)."""
        torch = self.torch
        nq, d = int(q.shape[0]), int(q.shape[1])
        cuts = [0]
        if nq > 2 * self.CALL_CHUNK:
            # (r6) stages cut where the search cuts its chunks (lvs_flat_search_keys: launches of 32 768 queries, the remainder
            # last): a stage of the remainder (copy: a few MB), one chunk, then everything else - no stage splits a chunk into
            # the smaller, slower launches, and every copy but the first tiny one runs under a search
            r = nq % self.CALL_CHUNK
            cuts += [r, r + self.CALL_CHUNK] if r >= 256 else [self.CALL_CHUNK]
        else:
            for share in self.CALL_PIPELINE[:-1]:
                cuts.append(min(nq, (int(nq * share) + cuts[-1] + 255) // 256 * 256))
        cuts.append(nq)
        stages = [(a, b) for a, b in zip(cuts[:-1], cuts[1:]) if b > a]
        tdt = torch.float16 if q.dtype == np.float16 else torch.float32
        cur = torch.cuda.current_stream(self.device)
        # every stage's device buffer exists before the first search is queued: its copy need not wait for that search
        bufs = [torch.empty((b - a, d), dtype=tdt, device=self.device) for a, b in stages]
        allocated = cur.record_event()
        if getattr(self, "_d2h_stream", None) is None:
            self._d2h_stream = torch.cuda.Stream(device=self.device)
        Dh = torch.empty((nq, k), dtype=torch.float32, pin_memory=True)
        Ih = torch.empty((nq, k), dtype=torch.int64, pin_memory=True)
        flags = torch.zeros((1,), dtype=torch.int32, device=self.device)
        mode = corpus.mode
        for (a, b), buf in zip(stages, bufs):
            c = np.ascontiguousarray(q[a:b], dtype=np.float16 if q.dtype == np.float16 else np.float32)
            self._h2d(c, out=buf, after=allocated)
            queries = self.pack(buf, mode, normalize=normalize, exp=exp, check="lazy")
            if queries.flags is not None:
                flags |= queries.flags
            keys = self.search_keys(corpus, queries, k, metric, id_offset=id_offset)
            Dd, Id = self.keys_to_result(keys, metric, None, score_exp=self.score_exp_of(corpus, queries))
            done = cur.record_event()
            with torch.cuda.stream(self._d2h_stream):
                self._d2h_stream.wait_event(done)
                Dh[a:b].copy_(Dd, non_blocking=True)
                Ih[a:b].copy_(Id, non_blocking=True)
            Dd.record_stream(self._d2h_stream)
            Id.record_stream(self._d2h_stream)
        fh = torch.empty((1,), dtype=torch.int32, pin_memory=True)
        fh.copy_(flags, non_blocking=True)
        cur.synchronize()
        self._d2h_stream.synchronize()
        return Dh.numpy(), Ih.numpy(), int(fh[0])

    # ---- packing ----
    SCALE_TARGET_EXP = 6  # exp="auto": the largest |x| lands in [2^6, 2^7) - components ~10x smaller still have their lo half
                          # in fp16's normal range, values up to 500x larger than the sampled maximum still fit

    def absmax(self, x) -> float:
        """Largest |value| of a device tensor (float16 / float32) through ``lvs_absmax``."""
        torch = self.torch
        bits = torch.zeros((1,), dtype=torch.int32, device=self.device)
        t = x.contiguous()
        if t.dtype not in (torch.float16, torch.float32):
            t = t.to(torch.float32)
        n, d = (int(t.shape[0]), int(t.shape[1])) if t.dim() == 2 else (1, int(t.numel()))
        self._c("lvs_absmax", _ptr(t), _capi.DTYPE_F16 if t.dtype == torch.float16 else _capi.DTYPE_F32, n, d, _ptr(bits),
                self._stream())
        return float(np.array([int(bits.item())], np.uint32).view(np.float32)[0])

    @classmethod
    def exp_for(cls, absmax: float) -> int:
        """The pack exponent that moves ``absmax`` to [2^SCALE_TARGET_EXP, 2^(SCALE_TARGET_EXP + 1))."""
        if not np.isfinite(absmax) or absmax <= 0.0:
            return 0
        return int(np.clip(cls.SCALE_TARGET_EXP - int(np.floor(np.log2(absmax))), -60, 60))

    def pack(self, x, mode: int, normalize: bool = False, check=False, exp=0) -> PackedRows:
        """x: numpy [n,d] (float16/32/64) or a torch CUDA tensor (float16/float32).

        ``exp``: every value is multiplied by 2^exp before it is rounded (exact).  ``"auto"`` picks the exponent from the
        data (hi|lo rows only: fp32-accurate rows keep ~22 significant bits whatever the embeddings' magnitude; fp16 rows
        are stored as given): from the first chunk's largest magnitude, with 500x headroom - should a later chunk exceed
        it, everything is packed again with the exponent of the true maximum.
        ``check``: validate the values on the device while packing.  ``True`` raises ``ValueError`` for inf / NaN or for
        magnitudes beyond fp16's range (one read-back of a flag word; faiss itself accepts any float32, the fp16-based
        rows here do not - DESIGN.md section 7); ``"lazy"`` leaves the flag word on the device (``PackedRows.flags``) for
        the caller to read together with its results (``raise_for_flags``)."""
        torch = self.torch
        is_tensor = torch.is_tensor(x)
        n, d = int(x.shape[0]), int(x.shape[1])
        ld = int(self.lib.lvs_packed_ld(d, mode))
        if ld <= 0:
            raise LotusHipError(f"bad dimension d={d}")
        auto = isinstance(exp, str)
        if auto and exp != "auto":
            raise ValueError("exp must be an int or 'auto'")
        if auto and normalize and mode == _capi.PACK_SPLIT:
            auto, exp = False, self.SCALE_TARGET_EXP + 2  # unit rows: components <= 1, typically ~3 / sqrt(d)
        if auto and (mode != _capi.PACK_SPLIT or n == 0):
            auto, exp = False, 0
        rows = torch.empty((n, ld), dtype=torch.float16, device=self.device)
        norms = torch.empty((n,), dtype=torch.float32, device=self.device)
        flags = torch.zeros((1,), dtype=torch.int32, device=self.device) if (check or auto) else None
        amax_bits = torch.zeros((1,), dtype=torch.int32, device=self.device) if auto else None
        step = self.PACK_CHUNK_ROWS

        def chunk_of(r0, r1):
            if is_tensor:
                chunk = x[r0:r1].contiguous()
                if chunk.dtype not in (torch.float16, torch.float32):
                    chunk = chunk.to(torch.float32)
                if chunk.device != self.device:
                    chunk = chunk.to(self.device)
                return chunk
            c = x[r0:r1]
            c = np.ascontiguousarray(c, dtype=np.float16 if c.dtype == np.float16 else np.float32)
            return self._h2d(c)

        def run(e: int):
            for r0 in range(0, n, step):
                r1 = min(n, r0 + step)
                chunk = chunk_of(r0, r1)
                src_dtype = _capi.DTYPE_F16 if chunk.dtype == torch.float16 else _capi.DTYPE_F32
                if auto:
                    self._c("lvs_absmax", _ptr(chunk), src_dtype, r1 - r0, d, _ptr(amax_bits), self._stream())
                    if r0 == 0 and e is None:  # the exponent comes from the first chunk (one read-back per index build)
                        e = self.exp_for(float(np.array([int(amax_bits.item())], np.uint32).view(np.float32)[0]))
                self._c("lvs_pack_rows_checked", _ptr(chunk), src_dtype, r1 - r0, d, mode, int(bool(normalize)), int(e),
                        _ptr(rows[r0:r1]), _ptr(norms[r0:r1]), _ptr(flags), self._stream())
                del chunk
            return e

        e = run(None if auto else int(exp))
        if auto and n > step:  # a later chunk may have exceeded the first chunk's range: pack again with the true maximum's
            f = int(flags.item())
            if f & _capi.PACK_FLAG_RANGE and not f & _capi.PACK_FLAG_NONFINITE:
                flags.zero_()
                e = run(self.exp_for(float(np.array([int(amax_bits.item())], np.uint32).view(np.float32)[0])))
        out = PackedRows(rows=rows, norms=norms, n=n, d=d, mode=mode, exp=int(e or 0), flags=flags if check == "lazy" else None)
        if check is True or (auto and check != "lazy"):
            self.raise_for_flags(int(flags.item()))
        return out

    @staticmethod
    def raise_for_flags(f: int, what: str = "embeddings") -> None:
        if f & _capi.PACK_FLAG_NONFINITE:
            raise ValueError(f"{what} contain inf or NaN")
        if f & _capi.PACK_FLAG_RANGE:
            raise ValueError(f"{what}: values exceed fp16's range (|x| > 65504 after the index's power-of-two scale): "
                             "rescale them (the device rows are fp16 or fp16 hi|lo pairs)")

    def gather(self, src: PackedRows, ids_dev) -> PackedRows:
        torch = self.torch
        m = int(ids_dev.numel())
        ld = int(src.rows.shape[1])
        rows = torch.empty((m, ld), dtype=torch.float16, device=self.device)
        norms = torch.empty((m,), dtype=torch.float32, device=self.device)
        self._c("lvs_gather_rows", _ptr(src.rows), ld, _ptr(ids_dev), m, _ptr(rows), self._stream())
        self._c("lvs_gather_f32", _ptr(src.norms), _ptr(ids_dev), m, _ptr(norms), self._stream())
        return PackedRows(rows=rows, norms=norms, n=m, d=src.d, mode=src.mode, exp=src.exp)

    def unpack(self, src: PackedRows, ids_dev=None, raw: bool = False):
        """float32 values [m, d] of the packed rows ``ids_dev`` (all rows when None), as a device tensor - the caller's
        values (pack scale undone); ``raw=True``: the stored values x 2^exp as they are (k-means works on those)."""
        torch = self.torch
        m = src.n if ids_dev is None else int(ids_dev.numel())
        out = torch.empty((m, src.d), dtype=torch.float32, device=self.device)
        self._c("lvs_unpack_rows", _ptr(src.rows), src.d, src.mode, _ptr(ids_dev), m, 0 if raw else int(src.exp), _ptr(out),
                self._stream())
        return out

    @staticmethod
    def slice_rows(src: PackedRows, r0: int, r1: int) -> PackedRows:
        """Rows [r0, r1) of a packed matrix (a view: same HBM)."""
        return PackedRows(rows=src.rows[r0:r1], norms=src.norms[r0:r1], n=r1 - r0, d=src.d, mode=src.mode, exp=src.exp)

    # ---- search ----
    CERT_MIN_PAIRS = 1 << 26  # below this many (query, row) pairs the extra launches cost more than the two passes saved
    CERT_MAX_K = 48
    CERT_SPARE_SMALL_K = 5  # spare list slots of the one-pass search for k <= 10 (k1 = k + spare <= 15)
    # largest k served by the 15-slot lists of the 256-query geometry.  fp32 10 k x 1 M, same box (profiles/r06l_fp32_k12_probe.log):
    # k = 11: 15.9 ms (0.4 % of the queries open) against 20.5 ms with k + 8 slots on the 128-query geometry; k = 12: 17.0 (3.8 % open)
    # against 20.6; k = 13: 21.7 (17 % open) against 20.7
    CERT_MAX_K_SMALL_LISTS = 12
    CERT_RJ_MIN_QUERIES = 4097  # from here on the certified search's one pass takes the chunked lvs_rj_kernel path (plain lists)
    CERT_BAND_SEARCH = 2.05   # banded lists: rows further than this many error bounds below the k-th one-pass score are not
    CERT_BAND_CERTIFY = 2.02  # listed; the certificate assumes a slightly narrower band (> 2 is what its proof needs)

    SEED_EXCHANGE_MIN_QUERIES = 2048  # below this a sharded call is too short for an extra collective (see seed_tiles)

    def seed_tiles(self, nq: int, shard_rows: int, k: int, corpus_mode: int, query_mode: int) -> int:
        """Sample tiles every shard of a row-sharded join contributes to the pooled starting thresholds (0: no exchange).
        A function of the arguments only - every rank computes it from the NOMINAL shard size, so all ranks agree on the
        shape of the exchange.  fp32-accurate operands (hi|lo rows) are searched through the certified one-pass path, whose
        certificate reads a short list as "this shard holds no more rows": no pooled thresholds there."""
        if nq < self.SEED_EXCHANGE_MIN_QUERIES or corpus_mode != _capi.PACK_F16 or query_mode != _capi.PACK_F16:
            return 0
        if not 1 <= k <= 56:
            return 0
        return max(0, int(self.lib.lvs_flat_search_seed_tiles(int(nq), int(shard_rows), int(k))))

    def seed_scores(self, corpus: PackedRows, queries: PackedRows, metric: int, tiles: int):
        """-> float32 [tiles, nq]: best score of every query over each of the first ``tiles`` 256-row tiles of this shard
        (``lvs_flat_search_seed_scores``; -inf rows where the shard is shorter)."""
        torch = self.torch
        out = torch.empty((int(tiles), queries.n), dtype=torch.float32, device=self.device)
        if tiles and queries.n:
            self._c("lvs_flat_search_seed_scores", _ptr(corpus.rows), corpus.mode, corpus.n, _ptr(queries.rows), queries.mode,
                    queries.n, corpus.d, metric, _ptr(corpus.norms), _ptr(queries.norms), int(tiles), _ptr(out),
                    self._stream())
        return out

    def search_keys(self, corpus: PackedRows, queries: PackedRows, k: int, metric: int, id_offset: int = 0,
                    row_ids=None, one_pass: bool | None = None, stats: dict | None = None, seed_scores=None):
        """-> int64 tensor [nq, k] holding the uint64 result keys (bit pattern).

        ``seed_scores`` (float32 [rows, nq], optional): scores of rows of the WHOLE searched set - the all-gathered
        ``seed_scores`` blocks of every corpus shard - whose k-th largest per query becomes the launch's starting threshold
        (``lvs_flat_search_keys_seeded``); a shard may then return fewer than k keys (key 0 slots).

        fp32-accurate operands (fp16 hi|lo rows) need 2-3 MFMA passes in the plain search.  For large calls with
        k <= 48 the same exact result comes from ONE pass (``one_pass``: None = decide by size, False = never):
        ``lvs_flat_search_keys_hi`` with a few spare list slots, exact rescoring of the candidates, a per-query
        certificate (``lvs_certify_topk``) and the plain search for the few uncertified queries only."""
        torch = self.torch
        if corpus.d != queries.d:
            raise ValueError("corpus / query dimension mismatch")
        if metric == _capi.METRIC_L2 and corpus.exp != queries.exp:
            raise ValueError("squared L2 needs both operands packed with the same scale exponent")
        split = corpus.mode == _capi.PACK_SPLIT or queries.mode == _capi.PACK_SPLIT
        if one_pass is None:
            one_pass = queries.n * corpus.n >= self.CERT_MIN_PAIRS
        if (one_pass and split and row_ids is None and 1 <= k <= self.CERT_MAX_K and corpus.n >= 4 * (k + 8)
                and queries.n > 0):
            if k == 1:  # the winner + runner-up certificate is the cheaper one-pass form for a single neighbour
                return self.nearest(corpus, queries, metric, id_offset=id_offset, stats=stats)
            return self._search_keys_certified(corpus, queries, k, metric, id_offset, stats)
        keys = torch.empty((queries.n, k), dtype=torch.int64, device=self.device)
        need = int(self.lib.lvs_flat_search_workspace_bytes(queries.n, corpus.n, corpus.d, k, corpus.mode, queries.mode))
        if need < 0:
            raise LotusHipError("lvs_flat_search_workspace_bytes rejected the shape")
        ws = self._workspace(need)
        if seed_scores is not None and int(seed_scores.shape[0]) > 0:
            if seed_scores.dim() != 2 or int(seed_scores.shape[1]) != queries.n or seed_scores.dtype != torch.float32:
                raise ValueError("seed_scores must be float32 [rows, nq]")
            seed_scores = seed_scores.contiguous()
            self._c("lvs_flat_search_keys_seeded", _ptr(corpus.rows), corpus.mode, corpus.n, _ptr(queries.rows), queries.mode,
                    queries.n, corpus.d, metric, k, _ptr(corpus.norms), _ptr(queries.norms), int(id_offset), _ptr(row_ids),
                    _ptr(seed_scores), int(seed_scores.shape[0]), _ptr(keys), _ptr(ws), int(ws.numel()), self._stream())
            return keys
        self._c("lvs_flat_search_keys", _ptr(corpus.rows), corpus.mode, corpus.n, _ptr(queries.rows), queries.mode,
                queries.n, corpus.d, metric, k, _ptr(corpus.norms), _ptr(queries.norms), int(id_offset), _ptr(row_ids),
                _ptr(keys), _ptr(ws), int(ws.numel()), self._stream())
        return keys

    def lo_norm_max(self, pk: PackedRows) -> float:
        """Largest |lo part of a row| of a device image (0 for fp16 rows), computed once per image and cached on it."""
        if pk.mode != _capi.PACK_SPLIT or pk.n == 0:
            return 0.0
        v = getattr(pk, "_lo_norm_max", None)
        if v is None:
            torch = self.torch
            dpad = int(pk.rows.shape[1]) // 2
            m = torch.zeros((), dtype=torch.float32, device=self.device)
            for r0 in range(0, pk.n, 1 << 18):  # in chunks: the float32 copy of 2^18 x 768 lo parts is 0.8 GB
                m = torch.maximum(m, pk.rows[r0:r0 + (1 << 18), dpad:].float().square().sum(dim=1).max())
            v = float(m.sqrt().item())
            try:
                pk._lo_norm_max = v
            except Exception:
                pass
        return v

    def row_norm_max(self, pk: PackedRows) -> float:
        """Largest |row| of a device image, computed once per image and cached on it."""
        if pk.n == 0:
            return 0.0
        v = getattr(pk, "_row_norm_max", None)
        if v is None:
            v = float(pk.norms.max().sqrt().item())
            try:
                pk._row_norm_max = v
            except Exception:
                pass
        return v

    def _search_call(self, name: str, corpus, queries, k, metric, id_offset):
        torch = self.torch
        keys = torch.empty((queries.n, k), dtype=torch.int64, device=self.device)
        need = int(self.lib.lvs_flat_search_workspace_bytes(queries.n, corpus.n, corpus.d, k, corpus.mode, queries.mode))
        if need < 0:
            raise LotusHipError("lvs_flat_search_workspace_bytes rejected the shape")
        ws = self._workspace(need)
        self._c(name, _ptr(corpus.rows), corpus.mode, corpus.n, _ptr(queries.rows), queries.mode, queries.n, corpus.d, metric,
                k, _ptr(corpus.norms), _ptr(queries.norms), int(id_offset), None, _ptr(keys), _ptr(ws), int(ws.numel()),
                self._stream())
        return keys

    def _search_keys_certified(self, corpus, queries, k, metric, id_offset, stats, k1=None):
        """Exact top-k of fp32-accurate operands from one MFMA pass (see ``search_keys``).

        The one-pass score of a pair differs from the exact one by at most ``|q| |lo_row| + |lo_q| |row| <= 2^-11 |q| R``
        per split operand (``R`` = largest row norm; every component is rounded to 11 significant bits).  With k1 > k
        list slots, a row that is NOT among a query's k1 candidates has a one-pass score <= the list's last one, so its
        exact score is at most that + the bound; if the k-th EXACT score of the candidates is strictly higher, their
        exact top k is the exact top k."""
        torch = self.torch
        nq = queries.n
        first_round = k1 is None
        if first_round:
            # k <= 12 stays on the 256-query geometry (15 slots): banded lists make spare slots cheap, and the 128-query geometry
            # of deeper lists is 25-45 % slower - a few more open queries for the second round are the smaller price
            k1 = min(15, k + self.CERT_SPARE_SMALL_K) if k <= 10 else (15 if k <= self.CERT_MAX_K_SMALL_LISTS else min(56, k + 8))
        k1 = min(k1, corpus.n)
        # |s - s_hi| <= |q| |lo_row| + |lo_q| |row| + |lo_q| |lo_row| (Cauchy-Schwarz) with the MEASURED largest lo-part norms
        # E_c, E_q of the two operands (cached per device image; ~0.4 x the worst case 2^-11 |x|, so fewer queries stay open than
        # under round 3's a-priori bound; being measured, they also cover components in fp16's subnormal range), + float32
        # accumulation noise relative to |q| |row|
        R = self.row_norm_max(corpus)
        E_c, E_q = self.lo_norm_max(corpus), self.lo_norm_max(queries)
        c = 1.0 if metric == _capi.METRIC_IP else 2.0
        scale = c * (E_c * (1.0 + 1e-3) + 8e-6 * R)
        slack = 1e-6 * (2.0 ** (corpus.exp + queries.exp) + R * R) + c * E_q * (R + E_c) * (1.0 + 1e-3)
        # BANDED lists: with bound = scale |q| + slack, a row whose one-pass score is more than 2 bound below the k-th best
        # one-pass score cannot reach the exact top k (k-th exact >= k-th one-pass - bound).  So the kernel admits a row only
        # above max(last slot, k-th slot - CERT_BAND_SEARCH x bound): the k1-deep lists then cost the insertions of k-deep
        # ones, and a query stays open only when more than k1 - k rows crowd that band.  The certificate assumes the slightly
        # narrower CERT_BAND_CERTIFY (rounding of the two evaluations of the band can never make it claim too much).
        keys = torch.empty((nq, k1), dtype=torch.int64, device=self.device)
        need = int(self.lib.lvs_flat_search_workspace_bytes(nq, corpus.n, corpus.d, k1, corpus.mode, queries.mode))
        if need < 0:
            raise LotusHipError("lvs_flat_search_workspace_bytes rejected the shape")
        ws = self._workspace(need)
        banded = k1 > k
        # (r6) joins of fp32 embeddings: the one pass over the hi parts runs on the register-resident-queries kernel
        # (lvs_rj_kernel: chunks of <= 32 768 queries, 0.5 instead of 0.43 of the MFMA roof) whenever the shape is its - that kernel
        # has no banded admission, so the lists are plain k1-deep ones (k1 <= 16 there) and the certificate the plain one
        dpad = int(corpus.rows.shape[1]) // (2 if corpus.mode == _capi.PACK_SPLIT else 1)
        if (banded and first_round and k1 <= 16 and nq >= self.CERT_RJ_MIN_QUERIES and corpus.n >= 65536
                and dpad in (256, 384, 512, 768) and corpus.mode == queries.mode == _capi.PACK_SPLIT):
            banded = False
        if banded:
            self._c("lvs_flat_search_keys_hi_banded", _ptr(corpus.rows), corpus.mode, corpus.n, _ptr(queries.rows), queries.mode,
                    nq, corpus.d, metric, k1, k, float(self.CERT_BAND_SEARCH * scale), float(self.CERT_BAND_SEARCH * slack),
                    _ptr(corpus.norms), _ptr(queries.norms), int(id_offset), None, _ptr(keys), _ptr(ws), int(ws.numel()),
                    self._stream())
            approx = keys
        else:
            approx = self._search_call("lvs_flat_search_keys_hi", corpus, queries, k1, metric, id_offset)
        exact = approx.clone()
        self._c("lvs_rescore_keys", _ptr(corpus.rows), corpus.mode, _ptr(queries.rows), queries.mode, nq, corpus.d, metric,
                _ptr(corpus.norms), _ptr(queries.norms), int(id_offset), k1, _ptr(exact), self._stream())
        self._c("lvs_sort_keys_desc", _ptr(exact), nq, k1, self._stream())
        idx = torch.empty((nq,), dtype=torch.int64, device=self.device)
        cnt = torch.zeros((1,), dtype=torch.int64, device=self.device)
        self._c("lvs_certify_topk_banded", _ptr(approx), _ptr(exact), _ptr(queries.norms), nq, k1, k, float(scale), float(slack),
                float(self.CERT_BAND_CERTIFY if banded else 0.0), _ptr(idx), _ptr(cnt), self._stream())
        keys = exact[:, :k].contiguous()
        n_open = int(cnt.item())
        n_plain = 0
        if n_open:
            sel = idx[:n_open]
            sub = self.gather(queries, sel)
            if first_round and k1 < min(56, corpus.n):
                # second round for the few open queries: the same one-pass search with the longest lists (56 slots) - a row
                # outside a 56-deep list is far below the k-th exact score unless the query sits in a dense tie - before
                # anyone pays the 2-3 segment plain search (1 % of 10 k x 1 M fp32 queries: 1.2 ms of plain search -> 0.4 ms)
                inner = {}
                keys[sel] = self._search_keys_certified(corpus, sub, k, metric, id_offset, inner, k1=56)
                n_plain = inner.get("plain", 0)
            else:
                keys[sel] = self.search_keys(corpus, sub, k, metric, id_offset=id_offset, one_pass=False)
                n_plain = n_open
        if stats is not None:
            if first_round:
                stats["uncertified"] = stats.get("uncertified", 0) + n_open
                stats["queries"] = stats.get("queries", 0) + nq
            stats["plain"] = stats.get("plain", 0) + n_plain
        return keys

    def nearest(self, corpus: PackedRows, queries: PackedRows, metric: int, id_offset: int = 0, stats: dict | None = None,
                exact_scores: bool = True, corpus_stats=None, bounds=None):
        """Nearest corpus row of every query (k = 1) -> int64 key tensor [nq, 1], same winner as ``search_keys(.., 1, ..)``
        (two rows whose exact scores agree to within float32 rounding may come out in either order: a certified winner needs no
        second look, a PAIR is settled by ``lvs_resolve_pairs``' scalar float32 dot products, an OPEN query by the multi-segment
        MFMA search - three summation orders; ``stats`` counts ``pairs`` and ``open`` separately, ``uncertified`` is their sum).

        fp32-accurate operands (fp16 hi|lo rows) cost two or three MFMA passes in the exact search.  Here ONE pass over
        the hi parts (``lvs_nearest_hi``) gives every query a winner and its margin over the runner-up; the true score
        of a pair differs from the hi-only score by at most |q| |lo_row| + |lo_q| |row| (Cauchy-Schwarz), so a margin
        above twice that bound certifies the winner.  Only the uncertified queries (``lvs_margin_select``: ties,
        near-ties) are searched again exactly.  This is the k-means assignment step (``lotus/utils.py:62,65``) with
        fp32-accurate centroids at the cost of fp16 ones.  ``exact_scores=False`` leaves the one-pass scores inside the
        keys of certified queries (the ids are exact either way) and saves one pass over the queries.  ``corpus_stats``:
        device float32 [2] = (largest |row|^2, largest |lo part of a row|^2) of the corpus as ``kmeans_pack_centroids`` /
        ``kmeans_finish`` leave it - the certificate's bound is then evaluated on the device (no host round trip for it).
        ``bounds``: ``(assign int32, ub float32, lb float32, positions int64 or None)`` device tensors - squared-L2 searches
        only, needs ``corpus_stats``: the rows ``positions`` (None: rows 0..nq) also get their Hamerly bounds from this
        search (``lvs_kmeans_bounds_set``): the winner, an upper bound of its distance and a lower bound of the distance to
        every other corpus row, the search's own error bound included."""
        torch = self.torch
        if bounds is not None and (corpus_stats is None or metric != _capi.METRIC_L2):
            raise ValueError("bounds need the squared-L2 metric and the corpus statistics")
        if corpus.mode == _capi.PACK_F16 and queries.mode == _capi.PACK_F16 and bounds is None:
            return self.search_keys(corpus, queries, 1, metric, id_offset=id_offset)  # nothing to certify: already exact
        plain = dict(id_offset=id_offset, one_pass=False)
        if corpus.d != queries.d:
            raise ValueError("corpus / query dimension mismatch")
        nq = queries.n
        keys = torch.empty((nq, 1), dtype=torch.int64, device=self.device)
        if nq == 0 or corpus.n == 0:
            return self.search_keys(corpus, queries, 1, metric, **plain)
        # error bound per unit of |q|: E = largest lo-part norm and R = largest norm over the corpus rows (a small matrix in
        # the k-means use: the centroids).  bound = scale |q| + slack with
        #   scale = c (E + 2^-11 R [queries split] + 8e-6 R) + 2^-16 cm R     (c = 2 / 4, cm = 1 / 2 for IP / L2; 8e-6: fp32
        #   slack = 1e-6 (1 + R^2) + [L2] 2^-16 R^2 + [queries split] c sqrt(d) 2^-25 R      accumulation noise of both searches)
        # lvs_nearest_hi tags every running score in its low six mantissa bits (relative perturbation < 2^-17 of u = q.y or
        # 2 q.y - |y|^2, for the winner and for the runner-up): the 2^-16 terms.  |lo_q| <= 2^-11 |q| + sqrt(d) 2^-25: the
        # second term covers components in fp16's subnormal range.
        coef, dpad = self._nearest_coef(corpus, queries, metric)
        if corpus.n <= _capi.NEAREST3_MAX_ROWS:
            # a small corpus (k-means centroids): queries streaming past resident corpus tiles + the two-candidate certificate.
            # Its scratch is 20 B per (corpus tile, query): the queries go through in chunks that keep it below a fixed budget
            # (10 M rows x 16 384 centroids would ask for 12.8 GB in one call - ADVICE r04); results are per query, so
            # chunking changes nothing
            step = self._nearest3_step(corpus)
            if nq <= step:
                return self._nearest3(corpus, queries, metric, id_offset, stats, exact_scores, corpus_stats, bounds, coef, dpad, plain)
            for q0 in range(0, nq, step):
                q1 = min(nq, q0 + step)
                bsub = bounds
                if bounds is not None:
                    pos = bounds[3][q0:q1] if bounds[3] is not None else torch.arange(q0, q1, dtype=torch.int64, device=self.device)
                    bsub = (bounds[0], bounds[1], bounds[2], pos)
                keys[q0:q1] = self._nearest3(corpus, self.slice_rows(queries, q0, q1), metric, id_offset, stats, exact_scores,
                                             corpus_stats, bsub, coef, dpad, plain)
            return keys
        if corpus_stats is None:
            R = float(corpus.norms.max().sqrt().item())
            E = self.lo_norm_max(corpus)
            scale = coef[0] * E + coef[1] * R
            slack = coef[2] + coef[3] * R + coef[4] * R * R
        sec = torch.empty((nq,), dtype=torch.float32, device=self.device)
        need = int(self.lib.lvs_nearest_hi_workspace_bytes(nq, corpus.n, corpus.d))
        ws = self._workspace(need)
        self._c("lvs_nearest_hi", _ptr(corpus.rows), corpus.mode, corpus.n, _ptr(queries.rows), queries.mode, nq, corpus.d,
                metric, _ptr(corpus.norms), _ptr(queries.norms), int(id_offset), _ptr(keys), _ptr(sec), _ptr(ws),
                int(ws.numel()), self._stream())
        idx = torch.empty((nq,), dtype=torch.int64, device=self.device)
        cnt = torch.zeros((1,), dtype=torch.int64, device=self.device)
        if corpus_stats is None:
            self._c("lvs_margin_select", _ptr(keys), _ptr(sec), _ptr(queries.norms), nq, float(scale), float(slack),
                    _ptr(idx), _ptr(cnt), self._stream())
        else:
            coef5 = (ctypes.c_float * 5)(*coef)
            self._c("lvs_margin_select_stats", _ptr(keys), _ptr(sec), _ptr(queries.norms), nq, _ptr(corpus_stats),
                    ctypes.addressof(coef5), _ptr(idx), _ptr(cnt), self._stream())
        if bounds is not None:  # every row's bounds from the one-pass result first; the uncertified rows are redone below
            b_assign, b_ub, b_lb, b_pos = bounds
            coef5 = (ctypes.c_float * 5)(*coef)
            self._c("lvs_kmeans_bounds_set", _ptr(keys), 1, _ptr(sec), _ptr(queries.norms), _ptr(b_pos), nq,
                    _ptr(corpus_stats), ctypes.addressof(coef5), int(id_offset), _ptr(b_assign), _ptr(b_ub), _ptr(b_lb),
                    self._stream())
        n_open = int(cnt.item())  # the call's one host round trip: the exact search below is sized by it
        # the winners are certified, their scores are still the one-pass approximations: put the exact scores in
        # (one HBM-bound pass over the queries; the k-means objective sums them)
        if exact_scores:
            self._c("lvs_rescore_keys", _ptr(corpus.rows), corpus.mode, _ptr(queries.rows), queries.mode, nq, corpus.d,
                    metric, _ptr(corpus.norms), _ptr(queries.norms), int(id_offset), 1, _ptr(keys), self._stream())
        if n_open:
            sel = idx[:n_open]
            sub = self.gather(queries, sel)
            exact_keys = self.search_keys(corpus, sub, 1, metric, **plain)  # the same exact search with or without bounds
            if bounds is not None:
                # float32 rounding of |x|^2 + |c|^2 - 2 x.c (cancellation when the row sits on a centroid): a few ulps of
                # the largest term - 8e-6 R |x| + 4e-6 R^2; errors proportional to the distance itself are covered by the
                # relative margin of lvs_kmeans_bounds_step
                exact2 = (ctypes.c_float * 2)(8e-6, 4e-6)
                pos = sel if b_pos is None else b_pos[sel]
                self._c("lvs_kmeans_bounds_fix", _ptr(keys[sel].contiguous()), _ptr(exact_keys), _ptr(sub.norms), _ptr(pos),
                        n_open, _ptr(corpus_stats), ctypes.addressof(coef5), ctypes.addressof(exact2), int(id_offset),
                        _ptr(b_assign), _ptr(b_ub), _ptr(b_lb), self._stream())
            keys[sel] = exact_keys
        if stats is not None:
            stats["uncertified"] = stats.get("uncertified", 0) + n_open
            stats["queries"] = stats.get("queries", 0) + nq
        return keys

    @staticmethod
    def _nearest_coef(corpus, queries, metric):
        """-> (the five coefficients of the one-pass error bound - see the comment in ``nearest`` -, dpad)."""
        dpad = int(corpus.rows.shape[1]) // (2 if corpus.mode == _capi.PACK_SPLIT else 1)
        qsplit = 1.0 if queries.mode == _capi.PACK_SPLIT else 0.0
        ip = metric == _capi.METRIC_IP
        c = 2.0 if ip else 4.0
        coef = [c, c * ((2.0 ** -11) * qsplit + 8e-6) + 2.0 ** -16 * (1.0 if ip else 2.0),
                1e-6 * 2.0 ** (corpus.exp + queries.exp), qsplit * c * (dpad ** 0.5) * 2.0 ** -25,
                1e-6 + (0.0 if ip else 2.0 ** -16)]
        return coef, dpad

    def _nearest3(self, corpus, queries, metric, id_offset, stats, exact_scores, corpus_stats, bounds, coef, dpad, plain):
        """``nearest`` for a corpus of at most 16 384 rows (``lvs_nearest3``): the one-pass search also returns the runner-up's
        id and the THIRD score, so a query whose best-second margin is inside the error bound but whose best-third margin is
        not has only two possible winners - two exact dot products (``lvs_resolve_pairs``) decide it instead of an exact
        search over every row; only queries with three or more rows inside the bound take that search.  The three-way split
        depends on a query's own scores and the corpus alone, never on which other queries share the call."""
        h = self._nearest3_begin(corpus, queries, metric, id_offset, exact_scores, corpus_stats, bounds, coef, dpad, plain)
        return self._nearest3_finish(h, stats)

    def _nearest3_begin(self, corpus, queries, metric, id_offset, exact_scores, corpus_stats, bounds, coef, dpad, plain):
        """Launches only: the one-pass search and the three-way split; nothing is read back.  -> handle for ``_nearest3_finish``
        (which may run on another stream once these launches are done - ``nearest_begin`` / ``nearest_finish``)."""
        torch = self.torch
        nq = queries.n
        if corpus_stats is None:  # (largest |row|^2, largest |lo part|^2) of the corpus, left on the device
            E2 = (corpus.rows[:, dpad:].float().square().sum(dim=1).max() if corpus.mode == _capi.PACK_SPLIT
                  else torch.zeros((), dtype=torch.float32, device=self.device))
            corpus_stats = torch.stack([corpus.norms.max(), E2]).to(torch.float32)
        keys = torch.empty((nq, 1), dtype=torch.int64, device=self.device)
        keys2 = torch.empty((nq,), dtype=torch.int64, device=self.device)
        sec = torch.empty((nq,), dtype=torch.float32, device=self.device)
        third = torch.empty((nq,), dtype=torch.float32, device=self.device)
        need = int(self.lib.lvs_nearest3_workspace_bytes(nq, corpus.n, corpus.d))
        if need < 0:
            raise LotusHipError("lvs_nearest3_workspace_bytes rejected the shape")
        ws = self._workspace(need)
        self._c("lvs_nearest3", _ptr(corpus.rows), corpus.mode, corpus.n, _ptr(queries.rows), queries.mode, nq, corpus.d, metric,
                _ptr(corpus.norms), _ptr(queries.norms), int(id_offset), _ptr(keys), _ptr(keys2), _ptr(sec), _ptr(third),
                _ptr(ws), int(ws.numel()), self._stream())
        pair_idx = torch.empty((nq,), dtype=torch.int64, device=self.device)
        open_idx = torch.empty((nq,), dtype=torch.int64, device=self.device)
        counts = torch.zeros((2,), dtype=torch.int64, device=self.device)
        coef5 = (ctypes.c_float * 5)(*coef)
        self._c("lvs_nearest3_select", _ptr(keys), _ptr(keys2), _ptr(sec), _ptr(third), _ptr(queries.norms), nq,
                _ptr(corpus_stats), ctypes.addressof(coef5), _ptr(pair_idx), _ptr(open_idx), _ptr(counts), self._stream())
        if bounds is not None:  # every row's bounds from the one-pass result first; the uncertified rows are redone below
            b_assign, b_ub, b_lb, b_pos = bounds
            self._c("lvs_kmeans_bounds_set", _ptr(keys), 1, _ptr(sec), _ptr(queries.norms), _ptr(b_pos), nq,
                    _ptr(corpus_stats), ctypes.addressof(coef5), int(id_offset), _ptr(b_assign), _ptr(b_ub), _ptr(b_lb),
                    self._stream())
        return dict(corpus=corpus, queries=queries, metric=metric, id_offset=id_offset, exact_scores=exact_scores,
                    corpus_stats=corpus_stats, bounds=bounds, coef5=coef5, plain=plain, keys=keys, keys2=keys2, sec=sec,
                    third=third, pair_idx=pair_idx, open_idx=open_idx, counts=counts)

    def _nearest3_finish(self, h, stats):
        """The call's one host round trip (how many queries are pairs / open) and what follows from it: two exact dot products
        for the pairs, the exact search for the open queries.  -> keys [nq, 1]."""
        corpus, queries, metric, id_offset = h["corpus"], h["queries"], h["metric"], h["id_offset"]
        corpus_stats, bounds, coef5, plain = h["corpus_stats"], h["bounds"], h["coef5"], h["plain"]
        keys, keys2, pair_idx, open_idx, counts = h["keys"], h["keys2"], h["pair_idx"], h["open_idx"], h["counts"]
        nq = queries.n
        if bounds is not None:
            b_assign, b_ub, b_lb, b_pos = bounds
        n_pair, n_open = (int(v) for v in counts.tolist())  # the call's one host round trip
        psel = pair_idx[:n_pair]
        # the one-pass keys of the rows decided below, as lvs_kmeans_bounds_fix needs them (taken before any exact score goes in)
        approx_pair = keys[psel].contiguous() if (bounds is not None and n_pair) else None
        approx_open = keys[open_idx[:n_open]].contiguous() if (bounds is not None and n_open) else None
        if h["exact_scores"]:
            self._c("lvs_rescore_keys", _ptr(corpus.rows), corpus.mode, _ptr(queries.rows), queries.mode, nq, corpus.d,
                    metric, _ptr(corpus.norms), _ptr(queries.norms), int(id_offset), 1, _ptr(keys), self._stream())
        exact2 = (ctypes.c_float * 2)(8e-6, 4e-6)  # float32 rounding of an exact distance, see nearest()
        if n_pair:
            self._c("lvs_resolve_pairs", _ptr(corpus.rows), corpus.mode, _ptr(queries.rows), queries.mode, corpus.d, metric,
                    _ptr(corpus.norms), _ptr(queries.norms), int(id_offset), _ptr(pair_idx), _ptr(counts), n_pair, _ptr(keys),
                    _ptr(keys2), self._stream())
            if bounds is not None:
                pos = psel if b_pos is None else b_pos[psel]
                self._c("lvs_kmeans_bounds_fix", _ptr(approx_pair), _ptr(keys[psel].contiguous()),
                        _ptr(queries.norms[psel].contiguous()), _ptr(pos), n_pair, _ptr(corpus_stats), ctypes.addressof(coef5),
                        ctypes.addressof(exact2), int(id_offset), _ptr(b_assign), _ptr(b_ub), _ptr(b_lb), self._stream())
        if n_open:
            sel = open_idx[:n_open]
            sub = self.gather(queries, sel)
            exact_keys = self.search_keys(corpus, sub, 1, metric, **plain)
            if bounds is not None:
                pos = sel if b_pos is None else b_pos[sel]
                self._c("lvs_kmeans_bounds_fix", _ptr(approx_open), _ptr(exact_keys), _ptr(sub.norms), _ptr(pos),
                        n_open, _ptr(corpus_stats), ctypes.addressof(coef5), ctypes.addressof(exact2), int(id_offset),
                        _ptr(b_assign), _ptr(b_ub), _ptr(b_lb), self._stream())
            keys[sel] = exact_keys
        if stats is not None:
            stats["uncertified"] = stats.get("uncertified", 0) + n_open + n_pair
            stats["pairs"] = stats.get("pairs", 0) + n_pair
            stats["open"] = stats.get("open", 0) + n_open
            stats["queries"] = stats.get("queries", 0) + nq
        return keys

    def nearest_begin(self, corpus: PackedRows, queries: PackedRows, metric: int, id_offset: int = 0, exact_scores: bool = True,
                      corpus_stats=None):
        """``nearest`` in two halves, for callers that keep the launch stream busy: ``nearest_begin`` queues the one-pass search and
        the certificate's three-way split and returns at once (no host round trip); ``nearest_finish(handle)`` - on this or on
        another stream that waits for an event recorded after ``nearest_begin`` - reads the two counts and queues the exact work
        for the uncertified queries.  Same launches, same arithmetic, same keys as ``nearest``.  Keep the handle alive until the
        stream ``nearest_finish`` ran on has been joined (its tensors were allocated on the stream of ``nearest_begin``)."""
        small = corpus.n <= _capi.NEAREST3_MAX_ROWS and not (corpus.mode == _capi.PACK_F16 and queries.mode == _capi.PACK_F16)
        if not small or queries.n == 0 or corpus.n == 0 or corpus.d != queries.d:
            return {"keys_now": self.nearest(corpus, queries, metric, id_offset=id_offset, exact_scores=exact_scores,
                                             corpus_stats=corpus_stats), "queries": queries}
        coef, dpad = self._nearest_coef(corpus, queries, metric)
        plain = dict(id_offset=id_offset, one_pass=False)
        step = self._nearest3_step(corpus)
        if queries.n <= step:
            return self._nearest3_begin(corpus, queries, metric, id_offset, exact_scores, corpus_stats, None, coef, dpad, plain)
        # the same scratch budget as nearest() (a 3 M-row range x 16 384 centroids asked for 3.8 GB in one call):
        # the queries go through in chunks, one handle each; the launches of chunk i + 1 reuse the stream's scratch behind chunk
        # i's (same stream: ordered)
        chunks = []
        for q0 in range(0, queries.n, step):
            q1 = min(queries.n, q0 + step)
            chunks.append((q0, q1, self._nearest3_begin(corpus, self.slice_rows(queries, q0, q1), metric, id_offset, exact_scores,
                                                        corpus_stats, None, coef, dpad, plain)))
        return {"chunks": chunks, "queries": queries}

    def _nearest3_step(self, corpus) -> int:
        """Most queries one ``lvs_nearest3`` call may take inside the scratch budget (20 B per corpus tile and query)."""
        per_q = max(1, int(self.lib.lvs_nearest3_workspace_bytes(1 << 20, corpus.n, corpus.d)) >> 20)
        return max(1 << 16, (self.NEAREST3_WS_BUDGET // per_q) >> 16 << 16)

    def nearest_finish(self, handle, stats: dict | None = None):
        if "keys_now" in handle:
            if stats is not None:
                stats["queries"] = stats.get("queries", 0) + handle["queries"].n
            return handle["keys_now"]
        if "chunks" in handle:
            keys = self.torch.empty((handle["queries"].n, 1), dtype=self.torch.int64, device=self.device)
            for q0, q1, h in handle["chunks"]:
                keys[q0:q1] = self._nearest3_finish(h, stats)
            return keys
        return self._nearest3_finish(handle, stats)

    def kmeans_centroid_shift(self, c_old, c_new):
        """-> (delta float32 [k], top2 float32 [3]): how far every centroid moved, the largest and second largest shift."""
        torch = self.torch
        k, d = int(c_new.shape[0]), int(c_new.shape[1])
        delta = torch.empty((k,), dtype=torch.float32, device=self.device)
        top2 = torch.empty((3,), dtype=torch.float32, device=self.device)
        self._c("lvs_kmeans_centroid_shift", _ptr(c_old), _ptr(c_new), k, d, _ptr(delta), _ptr(top2), self._stream())
        return delta, top2

    def kmeans_bounds_step(self, assign, ub, lb, delta, top2):
        """Move the bounds by the centroid shifts; -> int64 device tensor of the rows that need a new search."""
        torch = self.torch
        n = int(assign.numel())
        idx = torch.empty((n,), dtype=torch.int64, device=self.device)
        cnt = torch.zeros((1,), dtype=torch.int64, device=self.device)
        self._c("lvs_kmeans_bounds_step", _ptr(assign), _ptr(ub), _ptr(lb), _ptr(delta), _ptr(top2), n, _ptr(idx), _ptr(cnt),
                self._stream())
        return idx[:int(cnt.item())]

    def search_sharded(self, shard: PackedRows, queries: PackedRows, k: int, metric: int, id_offset: int, world: int,
                       seed_tiles: int, all_gather):
        """This rank's share of a row-sharded search with the exchange steps run INSIDE the C ABI (``lvs_search_sharded``, ABI 6):
        sample scores -> all-gather -> seeded shard search -> all-gather of the key lists -> merge, all enqueued on the current
        stream.  ``all_gather``: callable(device uint8 tensor [n]) -> device uint8 tensor [world, n] (rank order), called by the
        library for its two exchanges - ``lotus_amd._dist.all_gather_rows`` over any ``torch.distributed`` backend.  (A host with
        an ``ncclComm_t`` of its own calls ``lvs_search_sharded_rccl`` and needs no callback at all.)  -> int64 keys [nq, k],
        identical on every rank and equal to what ``search_keys`` + ``merge_keys`` give."""
        torch = self.torch
        nq = queries.n
        if shard.d != queries.d:
            raise ValueError("corpus / query dimension mismatch")
        errors = []
        FN = ctypes.CFUNCTYPE(ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p)

        def transport(_ctx, send, recv, nbytes, _stream):
            try:  # the library's stream is torch's current stream (self._stream()): torch work queued here stays in order
                src = torch.as_tensor(_DevBytes(send, nbytes), device=self.device)
                out = all_gather(src)
                torch.as_tensor(_DevBytes(recv, world * nbytes), device=self.device).copy_(out.reshape(-1))
                return 0
            except Exception as e:  # noqa: BLE001 - reported through the status code
                errors.append(e)
                return _capi.EDEVICE

        cb = FN(transport)
        need = int(self.lib.lvs_search_sharded_workspace_bytes(world, nq, shard.n, shard.d, k, shard.mode, queries.mode,
                                                               int(seed_tiles)))
        if need < 0:
            raise LotusHipError("lvs_search_sharded_workspace_bytes rejected the shape")
        ws = self._workspace(need)
        keys = torch.zeros((nq, k), dtype=torch.int64, device=self.device)
        with torch.cuda.device(self.device):
            st = self.lib.lvs_search_sharded(ctypes.cast(cb, ctypes.c_void_p), None, world, _ptr(shard.rows) if shard.n else None,
                                             shard.mode, shard.n, _ptr(queries.rows), queries.mode, nq, shard.d, metric, k,
                                             _ptr(shard.norms) if shard.n else None, _ptr(queries.norms), int(id_offset),
                                             int(seed_tiles), _ptr(keys), _ptr(ws), int(ws.numel()), self._stream())
        if errors:
            raise errors[0]
        _capi.check(st, "lvs_search_sharded")
        return keys

    def merge_keys(self, parts):
        """parts int64 [P, nq, k] -> [nq, k]."""
        torch = self.torch
        P, nq, k = (int(s) for s in parts.shape)
        out = torch.empty((nq, k), dtype=torch.int64, device=self.device)
        self._c("lvs_merge_keys", _ptr(parts.contiguous()), P, nq, k, _ptr(out), self._stream())
        return out

    def keys_to_result(self, keys, metric: int, id_map=None, score_exp: int = 0):
        """keys -> (D float32, I int64) device tensors.  ``score_exp``: sum of the two operands' pack exponents - the scores
        inside the keys are 2^score_exp x the caller's (``score_exp_of``)."""
        torch = self.torch
        nq, k = int(keys.shape[0]), int(keys.shape[1])
        D = torch.empty((nq, k), dtype=torch.float32, device=self.device)
        I = torch.empty((nq, k), dtype=torch.int64, device=self.device)
        self._c("lvs_keys_to_result", _ptr(keys), nq, k, metric, _ptr(id_map), int(score_exp), _ptr(D), _ptr(I),
                self._stream())
        return D, I

    @staticmethod
    def score_exp_of(corpus: PackedRows, queries: PackedRows) -> int:
        return int(corpus.exp) + int(queries.exp)

    def scores(self, corpus: PackedRows, queries: PackedRows, metric: int):
        torch = self.torch
        out = torch.empty((queries.n, corpus.n), dtype=torch.float32, device=self.device)
        if metric == _capi.METRIC_L2 and corpus.exp != queries.exp:
            raise ValueError("squared L2 needs both operands packed with the same scale exponent")
        self._c("lvs_scores", _ptr(corpus.rows), corpus.mode, corpus.n, _ptr(queries.rows), queries.mode, queries.n,
                corpus.d, metric, _ptr(corpus.norms), _ptr(queries.norms), self.score_exp_of(corpus, queries), _ptr(out),
                corpus.n, self._stream())
        return out

    def rank_scores(self, sc, id_offset: int = 0):
        """Rank every float32 score row best-first: ``sc`` [nq, nb] (larger = better) -> int64 key tensor [nq, nb].
        The device sort handles < 2^32 scores per launch, so the queries go through in chunks."""
        torch = self.torch
        nq, nb = int(sc.shape[0]), int(sc.shape[1])
        keys = torch.empty((nq, nb), dtype=torch.int64, device=self.device)
        if nq == 0 or nb == 0:
            return keys
        step = max(1, min(nq, (2**32 - 2) // nb))
        for q0 in range(0, nq, step):
            q1 = min(nq, q0 + step)
            need = int(self.lib.lvs_sort_rows_workspace_bytes(q1 - q0, nb))
            if need < 0:
                raise LotusHipError("lvs_sort_rows_workspace_bytes rejected the shape")
            ws = self._workspace(need)
            self._c("lvs_sort_rows_desc", _ptr(sc[q0:q1]), q1 - q0, nb, int(sc.stride(0)), int(id_offset),
                    _ptr(keys[q0:q1]), _ptr(ws), int(ws.numel()), self._stream())
        return keys

    def rank_all(self, corpus: PackedRows, queries: PackedRows, metric: int, id_offset: int = 0):
        """Every corpus row ranked for every query (K = N callers): int64 key tensor [nq, nb], best first."""
        if corpus.n >= 2**32 - 2:
            raise ValueError("K = N ranking needs fewer than 2^32 - 2 rows")
        return self.rank_scores(self.scores(corpus, queries, metric), id_offset)

    # ---- threshold join (sem_dedup) ----
    RANGE_CHUNK_ROWS = 65536  # query rows per launch (a multiple of 256 x any rank count that divides it)

    def range_join(self, corpus: PackedRows, queries: PackedRows, threshold: float, metric: int = _capi.METRIC_IP,
                   q_row0: int = -1, id_offset: int = 0, stride: int = 1, phase: int = 0, capacity: int = 1 << 22):
        """All (query, corpus id, score) with score > threshold.  q_row0 >= 0: self-join, pairs with id > query row
        only.  Returns three device tensors (int64, int64, float32); order unspecified.

        The query rows go through in chunks, each with its own pair buffer: when a chunk finds more than `capacity`
        pairs only that chunk is run again with a buffer of the counted size (the kernel keeps counting past the
        capacity) - a 5 M-row self-join repeats at most 1/77 of its work, not all of it."""
        torch = self.torch
        step = max(256 * stride, self.RANGE_CHUNK_ROWS // (256 * stride) * (256 * stride))
        # (r6) several ranks and a join the register-resident kernel's RANGE mode takes (fp16 rows, inner product, a long corpus):
        # whole CHUNKS of query rows are dealt to the ranks instead of 256-query tiles inside every launch - each chunk then runs
        # on that kernel as on one GPU.  Snake order (0 .. w-1, w-1 .. 0, ...): a self-join's early chunks see more rows than its
        # late ones, and plain round-robin would hand rank 0 the heavier chunk of every round.  Every rank applies the same rule,
        # so the ranks' pair lists partition the result.
        deal_chunks = (stride > 1 and metric == _capi.METRIC_IP and corpus.mode == queries.mode == _capi.PACK_F16
                       and corpus.n >= 2 * self.RANGE_CHUNK_ROWS and queries.n >= 2 * self.RANGE_CHUNK_ROWS
                       and int(corpus.rows.shape[1]) in (256, 384, 512, 768))
        if deal_chunks:
            step = self.RANGE_CHUNK_ROWS
        outs = []
        cnt = torch.zeros((1,), dtype=torch.int64, device=self.device)
        oq = oj = os_ = None
        for ci, c0 in enumerate(range(0, max(queries.n, 1), step)):
            c1 = min(queries.n, c0 + step)
            if c1 <= c0:
                break
            if deal_chunks:
                rnd, pos = divmod(ci, stride)
                if (pos if rnd % 2 == 0 else stride - 1 - pos) != phase:
                    continue
            qs = self.slice_rows(queries, c0, c1)
            cap = capacity
            while True:
                if oq is None or oq.numel() < cap:
                    oq = torch.empty((cap,), dtype=torch.int64, device=self.device)
                    oj = torch.empty((cap,), dtype=torch.int64, device=self.device)
                    os_ = torch.empty((cap,), dtype=torch.float32, device=self.device)
                cnt.zero_()
                self._c("lvs_range_join", _ptr(corpus.rows), corpus.mode, corpus.n, _ptr(qs.rows), qs.mode, qs.n,
                        corpus.d, metric, _ptr(corpus.norms), _ptr(qs.norms), float(threshold),
                        self.score_exp_of(corpus, qs), int(q_row0 + c0) if q_row0 >= 0 else -1, int(id_offset),
                        1 if deal_chunks else int(stride), 0 if deal_chunks else int(phase), int(cap),
                        _ptr(oq), _ptr(oj), _ptr(os_), _ptr(cnt), self._stream())
                n = int(cnt.item())
                if n <= cap:
                    break
                cap = n  # counted but not stored: size the buffers and run this chunk again
            if n:
                outs.append((oq[:n] + c0, oj[:n].clone(), os_[:n].clone()))
        if not outs:
            e = torch.empty((0,), dtype=torch.int64, device=self.device)
            return e, e.clone(), torch.empty((0,), dtype=torch.float32, device=self.device)
        return (torch.cat([o[0] for o in outs]), torch.cat([o[1] for o in outs]), torch.cat([o[2] for o in outs]))

    # ---- k-means pieces ----
    def kmeans_accumulate(self, x: PackedRows, assign, k: int):
        """-> (sums float32 [k,d], counts float32 [k]) of the rows of x grouped by assign (int64 device tensor)."""
        torch = self.torch
        sums = torch.zeros((k, x.d), dtype=torch.float32, device=self.device)
        counts = torch.zeros((k,), dtype=torch.float32, device=self.device)
        need = int(self.lib.lvs_kmeans_accumulate_workspace_bytes(x.n, k))
        ws = self._workspace(need)
        self._c("lvs_kmeans_accumulate", _ptr(x.rows), x.n, x.d, x.mode, _ptr(assign), k, _ptr(sums), _ptr(counts),
                _ptr(ws), int(ws.numel()), self._stream())
        return sums, counts

    def kmeans_accumulate_keys(self, x: PackedRows, keys, k: int, id_offset: int = 0):
        """The same straight from the assignment search's result keys [n(,1)] (no decode pass in between)."""
        torch = self.torch
        sums = torch.zeros((k, x.d), dtype=torch.float32, device=self.device)
        counts = torch.zeros((k,), dtype=torch.float32, device=self.device)
        need = int(self.lib.lvs_kmeans_accumulate_workspace_bytes(x.n, k))
        ws = self._workspace(need)
        self._c("lvs_kmeans_accumulate_keys", _ptr(x.rows), x.n, x.d, x.mode, _ptr(keys), int(id_offset), k, _ptr(sums),
                _ptr(counts), _ptr(ws), int(ws.numel()), self._stream())
        return sums, counts

    def kmeans_accumulate_keys_into(self, x: PackedRows, keys, k: int, sums, counts, id_offset: int = 0, workspace=None) -> None:
        """Add the rows of ``x`` to ``sums`` / ``counts`` CONTINUING the in-row-order sums that are there (zeros before the first
        rows): handing the rows over in consecutive ranges gives bit for bit what one call over all rows gives.  ``workspace``: a
        uint8 device tensor of ``lvs_kmeans_accumulate_workspace_bytes`` for a call that runs on a side stream next to launches
        using the backend's own workspace."""
        need = int(self.lib.lvs_kmeans_accumulate_workspace_bytes(x.n, k))
        ws = workspace if workspace is not None else self._workspace(need)
        if int(ws.numel()) < need:
            raise ValueError("workspace too small")
        self._c("lvs_kmeans_accumulate_keys", _ptr(x.rows), x.n, x.d, x.mode, _ptr(keys), int(id_offset), k, _ptr(sums),
                _ptr(counts), _ptr(ws), int(ws.numel()), self._stream())

    def kmeans_objective(self, centroids, sums, counts, x2, out) -> None:
        """out[0] (device float64) = x2[0] - 2 sum_j <c_j, S_j> + sum_j n_j |c_j|^2: faiss's objective of the iteration
        (sum of the assignment distances) from the sums, with the centroids BEFORE the update; float64 on the device."""
        k, d = int(centroids.shape[0]), int(centroids.shape[1])
        need = int(self.lib.lvs_kmeans_objective_workspace_bytes(k))
        if getattr(self, "_obj_ws", None) is None or self._obj_ws.numel() < need:
            self._obj_ws = self.torch.empty(need, dtype=self.torch.uint8, device=self.device)
        self._c("lvs_kmeans_objective", _ptr(centroids), _ptr(sums), _ptr(counts), k, d, _ptr(x2), _ptr(out),
                _ptr(self._obj_ws), int(self._obj_ws.numel()), self._stream())

    def kmeans_pack_centroids(self, centroids, mode: int, exp: int = 0):
        """-> (PackedRows of the centroids, device float32 [2] = (largest |c|^2, largest |lo part|^2)).  ``centroids`` are
        in the points' scaled domain already (``exp``: the points' pack exponent, recorded on the result)."""
        torch = self.torch
        k, d = int(centroids.shape[0]), int(centroids.shape[1])
        ld = int(self.lib.lvs_packed_ld(d, mode))
        rows = torch.empty((k, ld), dtype=torch.float16, device=self.device)
        norms = torch.empty((k,), dtype=torch.float32, device=self.device)
        cstats = torch.empty((2,), dtype=torch.float32, device=self.device)
        self._c("lvs_kmeans_pack_centroids", _ptr(centroids), k, d, mode, _ptr(rows), _ptr(norms), _ptr(cstats),
                self._stream())
        return PackedRows(rows=rows, norms=norms, n=k, d=d, mode=mode, exp=int(exp)), cstats

    def kmeans_finish(self, sums, counts, centroids, n_train: int, mode: int, nsplit_out=None, exp: int = 0):
        """The rest of a faiss iteration after the sums, one C-ABI call, nothing read back: centroids (device float32
        [k,d], in place) = sums / counts where counts > 0 (compute_centroids), faiss's empty-cluster split on the device
        (``nsplit_out``: device int32 [1]), and the repacked centroids -> (PackedRows, stats) as ``kmeans_pack_centroids``."""
        torch = self.torch
        k, d = int(centroids.shape[0]), int(centroids.shape[1])
        ld = int(self.lib.lvs_packed_ld(d, mode))
        rows = torch.empty((k, ld), dtype=torch.float16, device=self.device)
        norms = torch.empty((k,), dtype=torch.float32, device=self.device)
        cstats = torch.empty((2,), dtype=torch.float32, device=self.device)
        self._c("lvs_kmeans_update_centroids", _ptr(sums), _ptr(counts), k, d, int(n_train), _ptr(centroids),
                _ptr(nsplit_out), mode, _ptr(rows), _ptr(norms), _ptr(cstats), self._stream())
        return PackedRows(rows=rows, norms=norms, n=k, d=d, mode=mode, exp=int(exp)), cstats

    def kmeans_iteration(self, x: PackedRows, x2, k: int, n_train_total: int, centroids, cpk: PackedRows, cstats, keys_out,
                         obj_out, nsplit_out=None, all_reduce=None, stats: dict | None = None) -> None:
        """ONE exhaustive Lloyd iteration as one C-ABI call (``lvs_kmeans_iteration``, ABI 7): the certified assignment of the
        rows ``x`` against the centroids, the in-row-order sums, the objective, [``all_reduce(tensor)``: the in-place sum over the
        ranks of a float32 / float64 device tensor], division + faiss's split + repack - ``centroids`` (float32 [k,d]), ``cpk``
        (their hi|lo image) and ``cstats`` are updated in place, ``keys_out`` [n] receives the assignment, ``obj_out`` (float64
        [1]) the objective, ``nsplit_out`` (int32 [1]) the split count.  Bit-identical to ``nearest`` + ``kmeans_accumulate_keys``
        + ``kmeans_objective`` + ``kmeans_finish``.  One host round trip inside (the open rows' count)."""
        torch = self.torch
        need = int(self.lib.lvs_kmeans_iteration_workspace_bytes(x.n, x.d, int(k), x.mode, cpk.mode))
        if need < 0:
            raise LotusHipError("lvs_kmeans_iteration_workspace_bytes rejected the shape")
        ws = self._workspace(need)
        cb, errors = None, []
        if all_reduce is not None:
            FN = ctypes.CFUNCTYPE(ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32, ctypes.c_void_p)

            def transport(_ctx, buf, count, dtype, _stream):
                try:  # the library's stream is torch's current stream: torch work queued here stays in order
                    dt, size = (torch.float32, 4) if dtype == 0 else (torch.float64, 8)
                    all_reduce(torch.as_tensor(_DevBytes(buf, int(count) * size), device=self.device).view(dt))
                    return 0
                except Exception as e:  # noqa: BLE001 - reported through the status code
                    errors.append(e)
                    return _capi.EDEVICE

            fn = FN(transport)
            cb = ctypes.cast(fn, ctypes.c_void_p)
        host_counts = (ctypes.c_int64 * 2)()
        try:
            self._c("lvs_kmeans_iteration", cb, None, _ptr(x.rows), x.mode, x.n, x.d, _ptr(x.norms), _ptr(x2),
                    int(x.exp) + int(cpk.exp), int(k), int(n_train_total), _ptr(centroids), cpk.mode, _ptr(cpk.rows), _ptr(cpk.norms),
                    _ptr(cstats), _ptr(keys_out), _ptr(obj_out), _ptr(nsplit_out), ctypes.addressof(host_counts), _ptr(ws),
                    int(ws.numel()), self._stream())
        except LotusHipError:
            if errors:
                raise errors[0]
            raise
        if stats is not None:
            n_pair, n_open = int(host_counts[0]), int(host_counts[1])
            stats["uncertified"] = stats.get("uncertified", 0) + n_pair + n_open
            stats["pairs"] = stats.get("pairs", 0) + n_pair
            stats["open"] = stats.get("open", 0) + n_open
            stats["queries"] = stats.get("queries", 0) + x.n

    def kmeans_update_centroids(self, sums, counts, centroids) -> None:
        """centroids (device float32 [k,d], in place) = sums / counts where counts > 0 (faiss compute_centroids) - the
        division alone, for callers that split empty clusters on the host (``split_clusters``)."""
        k, d = int(centroids.shape[0]), int(centroids.shape[1])
        self._c("lvs_kmeans_update_centroids", _ptr(sums), _ptr(counts), k, d, 0, _ptr(centroids), None, 0, None, None,
                None, self._stream())

    def rand_perm(self, n: int, seed: int, m: int | None = None) -> np.ndarray:
        """faiss ``rand_perm(n, seed)``; with ``m`` only its first ``m`` entries (O(m) host time instead of O(n))."""
        if m is not None and m < n:
            out = np.empty(m, np.int64)
            _capi.check(self.lib.lvs_rand_perm_prefix_host(n, seed, m, out.ctypes.data), "lvs_rand_perm_prefix_host")
            return out
        out = np.empty(n, np.int64)
        _capi.check(self.lib.lvs_rand_perm_host(n, seed, out.ctypes.data), "lvs_rand_perm_host")
        return out

    def split_clusters(self, n: int, hassign: np.ndarray, centroids: np.ndarray) -> int:
        assert hassign.dtype == np.float32 and centroids.dtype == np.float32 and centroids.flags.c_contiguous
        k, d = centroids.shape
        ns = ctypes.c_int32(0)
        _capi.check(self.lib.lvs_kmeans_split_clusters_host(d, k, n, hassign.ctypes.data, centroids.ctypes.data,
                                                            ctypes.byref(ns)), "lvs_kmeans_split_clusters_host")
        return int(ns.value)

    # ---- measurement ----
    def timing_enable(self, on: bool) -> None:
        _capi.check(self.lib.lvs_timing_enable(int(on)))

    KERNEL_NAMES = {0: "lvs_tile_kernel", 1: "lvs_stream_kernel", 2: "lvs_rq_kernel", 3: "lvs_rj_kernel"}

    def timing_read(self):
        """-> (total ms of the dominant kernel's launches, searches that timed at least one): total / searches = kernel time per
        search whether it ran as one launch or - beyond 4 096 queries on the register-resident kernels - as one per chunk."""
        t = self.timing_read_full()
        return t["total_ms"], t["calls"]

    def timing_read_full(self) -> dict:
        tot, cnt, calls, kern = ctypes.c_double(0), ctypes.c_int64(0), ctypes.c_int64(0), ctypes.c_int32(0)
        _capi.check(self.lib.lvs_timing_read_calls(ctypes.byref(tot), ctypes.byref(cnt), ctypes.byref(calls), ctypes.byref(kern)))
        return {"total_ms": float(tot.value), "launches": int(cnt.value), "calls": int(calls.value),
                "kernel": self.KERNEL_NAMES.get(int(kern.value), str(int(kern.value)))}
