"""CPU timing comparator: faiss's BLAS search path rebuilt (a) in C + OpenMP with an AVX-512 sgemm micro-kernel fused with
the k-best collector (``oracle/c/lvs_blas_twin.c`` -> ``flat_search_c``, the one ``bench.py`` times) and (b) on torch-CPU
(MKL sgemm + ``topk`` -> ``flat_search_blas``), all host cores.
TEST / MEASUREMENT INFRASTRUCTURE ONLY - used by ``bench.py``'s ``cpu_baseline`` leg and checked against
``oracle/flat.py`` in ``tests/test_oracle.py``.

What it follows: faiss ``IndexFlat::search`` for >= 20 queries (the branch ``lotus/vector_store/faiss_vs.py:67,75``
takes for every batched call): query blocks x database blocks, one sgemm per block pair, block scores fed to a k-best
collector.  faiss's own database block is 1024 rows; SURVEY.md 8(d) measured that such small
blocks plus a Python-level collector leave > 10x of the CPU on the table and asks for a "faiss-equivalent CPU comparator"
with larger database blocks for TIMING - this is it (``DB_BLOCK`` rows per sgemm, ``torch.topk`` as the collector, a
running k-best merged block by block).  It is labelled "port", never "faiss": the real wheel is not installable here.

Result order is (score best-first); among exactly equal scores ``torch.topk`` gives no id-order guarantee, so parity
claims are made with ``oracle/flat.py``, not with this module.
"""
from __future__ import annotations

import os

import numpy as np

QUERY_BLOCK = 4096   # faiss distance_compute_blas_query_bs
DB_BLOCK = 65536     # rows per sgemm (faiss: 1024); 4096 x 65536 x 4 B = 1 GB of block scores


def flat_search_blas(xb, xq, k: int, metric: int = 0, threads: int | None = None):
    """-> (D float32 [nq,k], I int64 [nq,k], threads used).  metric 0: inner product (descending), 1: squared L2."""
    import torch

    threads = int(threads or os.cpu_count() or 1)
    torch.set_num_threads(threads)
    xb_t = torch.from_numpy(np.ascontiguousarray(xb, dtype=np.float32))
    xq_t = torch.from_numpy(np.ascontiguousarray(xq, dtype=np.float32))
    nb, nq = xb_t.shape[0], xq_t.shape[0]
    k_eff = min(k, nb)
    D = torch.full((nq, k), float("-inf"), dtype=torch.float32)
    I = torch.full((nq, k), -1, dtype=torch.int64)
    bn = (xb_t * xb_t).sum(1) if metric == 1 else None
    for q0 in range(0, nq, QUERY_BLOCK):
        q = xq_t[q0:q0 + QUERY_BLOCK]
        qn = (q * q).sum(1, keepdim=True) if metric == 1 else None
        best_v = torch.full((q.shape[0], 0), 0.0)
        best_i = torch.zeros((q.shape[0], 0), dtype=torch.int64)
        for r0 in range(0, nb, DB_BLOCK):
            r1 = min(nb, r0 + DB_BLOCK)
            s = q @ xb_t[r0:r1].T  # one sgemm per block pair
            if metric == 1:  # |x|^2 + |y|^2 - 2<x,y>, clamped at 0; "better" = minus the distance
                s = -torch.clamp((qn + bn[r0:r1][None, :]) - 2.0 * s, min=0.0)
            v, i = torch.topk(s, min(k_eff, r1 - r0), dim=1)
            best_v = torch.cat([best_v, v], dim=1)
            best_i = torch.cat([best_i, i + r0], dim=1)
            if best_v.shape[1] > k_eff:  # running k-best of the blocks seen so far
                v, j = torch.topk(best_v, k_eff, dim=1)
                best_v, best_i = v, torch.gather(best_i, 1, j)
        order = torch.argsort(best_v, dim=1, descending=True, stable=True)
        D[q0:q0 + q.shape[0], :k_eff] = torch.gather(best_v, 1, order)
        I[q0:q0 + q.shape[0], :k_eff] = torch.gather(best_i, 1, order)
    D, I = D.numpy(), I.numpy()
    FLT_MAX = np.float32(3.4028234663852886e38)
    if metric == 1:
        D = np.where(I >= 0, -D, FLT_MAX).astype(np.float32)
    else:
        D = np.where(I >= 0, D, -FLT_MAX).astype(np.float32)
    return D, I, threads


# ---- the C + OpenMP comparator ----------------------------------------------------------------------------------------
import ctypes  # noqa: E402

_HERE = os.path.dirname(os.path.abspath(__file__))
_TWIN_PATH = os.path.join(_HERE, "liblvs_blas_twin.so")
_twin = None


def c_available() -> bool:
    return os.path.exists(_TWIN_PATH)


def flat_search_c(xb, xq, k: int, metric: int = 0, threads: int = 0):
    """-> (D float32 [nq,k], I int64 [nq,k], threads).  Same key order as the oracle (score best-first, id ascending)."""
    global _twin
    from .flat import _finish

    if _twin is None:
        lib = ctypes.CDLL(_TWIN_PATH)
        lib.twin_flat_search.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32,
                                         ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p, ctypes.c_int32]
        lib.twin_flat_search.restype = None
        _twin = lib
    xb = np.ascontiguousarray(xb, dtype=np.float32)
    xq = np.ascontiguousarray(xq, dtype=np.float32)
    keys = np.zeros((xq.shape[0], k), np.uint64)
    threads = int(threads or os.cpu_count() or 1)
    _twin.twin_flat_search(xb.ctypes.data, xb.shape[0], xq.ctypes.data, xq.shape[0], xb.shape[1], k, metric,
                           keys.ctypes.data, threads)
    D, I = _finish(keys, metric, None)
    return D, I, threads
