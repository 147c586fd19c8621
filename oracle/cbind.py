"""ctypes binding of the plain-C oracle twin (``oracle/c/lvs_oracle.c``).  Test infrastructure only."""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liblvs_oracle.so")
_lib = None


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "c", "lvs_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.run(["make", "-C", _HERE, "-B" if force else "-s"], check=True, capture_output=True)
    return _LIB_PATH


def _load():
    global _lib
    if _lib is None and os.path.exists(_LIB_PATH):
        lib = ctypes.CDLL(_LIB_PATH)
        i64, i32, vp = ctypes.c_int64, ctypes.c_int32, ctypes.c_void_p
        lib.oracle_topk_update.argtypes = [vp, i64, i64, i64, i64, i32, vp]
        lib.oracle_topk_update.restype = None
        lib.oracle_flat_search_naive.argtypes = [vp, i64, vp, i64, i32, i32, i32, vp]
        lib.oracle_flat_search_naive.restype = None
        lib.oracle_mt19937_raw.argtypes = [ctypes.c_uint32, i64, vp]
        lib.oracle_mt19937_raw.restype = None
        lib.oracle_rand_perm.argtypes = [i64, i64, vp]
        lib.oracle_rand_perm.restype = None
        lib.oracle_compute_centroids.argtypes = [vp, i64, i32, i32, vp, vp, vp]
        lib.oracle_compute_centroids.restype = None
        lib.oracle_split_clusters.argtypes = [i32, i32, i64, vp, vp]
        lib.oracle_split_clusters.restype = i32
        lib.oracle_num_threads.restype = i32
        _lib = lib
    return _lib


def available() -> bool:
    return _load() is not None


def num_threads() -> int:
    return int(_load().oracle_num_threads())


def topk_update(better: np.ndarray, id0: int, keys: np.ndarray) -> None:
    """keys [nq,k] uint64 (C-contiguous view allowed) updated in place with one score block."""
    assert better.dtype == np.float32 and better.flags.c_contiguous
    assert keys.dtype == np.uint64 and keys.flags.c_contiguous
    nq, nb = better.shape
    _load().oracle_topk_update(better.ctypes.data, nq, nb, nb, int(id0), keys.shape[1], keys.ctypes.data)


def flat_search_naive(xb: np.ndarray, xq: np.ndarray, k: int, metric: int) -> np.ndarray:
    xb = np.ascontiguousarray(xb, np.float32)
    xq = np.ascontiguousarray(xq, np.float32)
    keys = np.zeros((xq.shape[0], k), np.uint64)
    if k > 0 and xq.shape[0] > 0:
        _load().oracle_flat_search_naive(xb.ctypes.data, xb.shape[0], xq.ctypes.data, xq.shape[0], xb.shape[1], k,
                                         metric, keys.ctypes.data)
    return keys


def mt19937_raw(seed: int, n: int) -> np.ndarray:
    out = np.zeros(n, np.uint32)
    _load().oracle_mt19937_raw(seed & 0xFFFFFFFF, n, out.ctypes.data)
    return out


def rand_perm(n: int, seed: int) -> np.ndarray:
    perm = np.zeros(n, np.int64)
    _load().oracle_rand_perm(n, seed, perm.ctypes.data)
    return perm


def compute_centroids(x: np.ndarray, assign: np.ndarray, centroids: np.ndarray) -> np.ndarray:
    n, d = x.shape
    k = centroids.shape[0]
    hassign = np.zeros(k, np.float32)
    assert x.dtype == np.float32 and x.flags.c_contiguous and centroids.flags.c_contiguous
    assign = np.ascontiguousarray(assign, np.int64)
    _load().oracle_compute_centroids(x.ctypes.data, n, d, k, assign.ctypes.data, centroids.ctypes.data,
                                     hassign.ctypes.data)
    return hassign


def split_clusters(n: int, hassign: np.ndarray, centroids: np.ndarray) -> int:
    k, d = centroids.shape
    assert hassign.dtype == np.float32 and centroids.dtype == np.float32
    return int(_load().oracle_split_clusters(d, k, n, hassign.ctypes.data, centroids.ctypes.data))
