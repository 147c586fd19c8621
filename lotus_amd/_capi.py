"""ctypes binding of ``liblotus_hip.so`` (C ABI in ``include/lotus_hip.h``).

The binding is deliberately thin: every function takes raw device addresses (ints) and sizes.  Loading fails
loudly when the library has not been built; there is no fallback implementation."""
from __future__ import annotations

import ctypes
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "liblotus_hip.so")
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "lotus_hip.h")

OK, EINVAL, ENOMEM, EDEVICE, EUNSUPPORTED = 0, -1, -2, -3, -4
DTYPE_F32, DTYPE_F16 = 0, 1
METRIC_IP, METRIC_L2 = 0, 1
PACK_F16, PACK_SPLIT = 0, 1
MAX_K = 2048
NEAREST3_MAX_ROWS = 16384
ABI_VERSION = 7
BUILD_TUNING, BUILD_COUNT_EVENTS = 1, 2
PACK_FLAG_NONFINITE, PACK_FLAG_RANGE = 1, 2

_i32, _i64, _vp, _dbl = ctypes.c_int32, ctypes.c_int64, ctypes.c_void_p, ctypes.c_double

# name -> (restype, argtypes); must list every symbol declared in include/lotus_hip.h
SIGNATURES = {
    "lvs_abi_version": (_i32, []),
    "lvs_last_error": (ctypes.c_char_p, []),
    "lvs_build_flags": (_i32, []),
    "lvs_device_count": (_i32, [ctypes.POINTER(_i32)]),
    "lvs_device_info": (_i32, [_i32, ctypes.c_char_p, _i32, ctypes.POINTER(_i32), ctypes.POINTER(_i64)]),
    "lvs_packed_ld": (_i32, [_i32, _i32]),
    "lvs_pack_rows": (_i32, [_vp, _i32, _i64, _i32, _i32, _i32, _vp, _vp, _vp]),
    "lvs_pack_rows_checked": (_i32, [_vp, _i32, _i64, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp]),
    "lvs_absmax": (_i32, [_vp, _i32, _i64, _i32, _vp, _vp]),
    "lvs_gather_rows": (_i32, [_vp, _i32, _vp, _i64, _vp, _vp]),
    "lvs_gather_f32": (_i32, [_vp, _vp, _i64, _vp, _vp]),
    "lvs_unpack_rows": (_i32, [_vp, _i32, _i32, _vp, _i64, _i32, _vp, _vp]),
    "lvs_flat_search_workspace_bytes": (_i64, [_i64, _i64, _i32, _i32, _i32, _i32]),
    "lvs_flat_search_keys": (_i32, [_vp, _i32, _i64, _vp, _i32, _i64, _i32, _i32, _i32, _vp, _vp, _i64, _vp, _vp,
                                    _vp, _i64, _vp]),
    "lvs_flat_search_seed_tiles": (_i32, [_i64, _i64, _i32]),
    "lvs_flat_search_seed_scores": (_i32, [_vp, _i32, _i64, _vp, _i32, _i64, _i32, _i32, _vp, _vp, _i32, _vp, _vp]),
    "lvs_flat_search_keys_seeded": (_i32, [_vp, _i32, _i64, _vp, _i32, _i64, _i32, _i32, _i32, _vp, _vp, _i64, _vp, _vp, _i32,
                                           _vp, _vp, _i64, _vp]),
    "lvs_merge_keys": (_i32, [_vp, _i32, _i64, _i32, _vp, _vp]),
    "lvs_search_sharded_workspace_bytes": (_i64, [_i32, _i64, _i64, _i32, _i32, _i32, _i32, _i32]),
    "lvs_search_sharded": (_i32, [_vp, _vp, _i32, _vp, _i32, _i64, _vp, _i32, _i64, _i32, _i32, _i32, _vp, _vp, _i64, _i32, _vp,
                                  _vp, _i64, _vp]),
    "lvs_rccl_available": (_i32, []),
    "lvs_rccl_bind": (_i32, [_vp, _vp, _vp]),
    "lvs_search_sharded_rccl": (_i32, [_vp, _vp, _i32, _i64, _vp, _i32, _i64, _i32, _i32, _i32, _vp, _vp, _i64, _i32, _vp, _vp,
                                       _i64, _vp]),
    "lvs_keys_to_result": (_i32, [_vp, _i64, _i32, _i32, _vp, _i32, _vp, _vp, _vp]),
    "lvs_scores": (_i32, [_vp, _i32, _i64, _vp, _i32, _i64, _i32, _i32, _vp, _vp, _i32, _vp, _i64, _vp]),
    "lvs_sort_rows_workspace_bytes": (_i64, [_i64, _i64]),
    "lvs_sort_rows_desc": (_i32, [_vp, _i64, _i64, _i64, _i64, _vp, _vp, _i64, _vp]),
    "lvs_range_join": (_i32, [_vp, _i32, _i64, _vp, _i32, _i64, _i32, _i32, _vp, _vp, ctypes.c_float, _i32, _i64, _i64, _i32,
                              _i32, _i64, _vp, _vp, _vp, _vp, _vp]),
    "lvs_nearest_hi_workspace_bytes": (_i64, [_i64, _i64, _i32]),
    "lvs_nearest_hi": (_i32, [_vp, _i32, _i64, _vp, _i32, _i64, _i32, _i32, _vp, _vp, _i64, _vp, _vp, _vp, _i64, _vp]),
    "lvs_nearest3_workspace_bytes": (_i64, [_i64, _i64, _i32]),
    "lvs_nearest3": (_i32, [_vp, _i32, _i64, _vp, _i32, _i64, _i32, _i32, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _i64, _vp]),
    "lvs_nearest3_select": (_i32, [_vp, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp]),
    "lvs_resolve_pairs": (_i32, [_vp, _i32, _vp, _i32, _i32, _i32, _vp, _vp, _i64, _vp, _vp, _i64, _vp, _vp, _vp]),
    "lvs_rescore_keys": (_i32, [_vp, _i32, _vp, _i32, _i64, _i32, _i32, _vp, _vp, _i64, _i32, _vp, _vp]),
    "lvs_flat_search_keys_hi": (_i32, [_vp, _i32, _i64, _vp, _i32, _i64, _i32, _i32, _i32, _vp, _vp, _i64, _vp, _vp,
                                       _vp, _i64, _vp]),
    "lvs_flat_search_keys_hi_banded": (_i32, [_vp, _i32, _i64, _vp, _i32, _i64, _i32, _i32, _i32, _i32, ctypes.c_float,
                                              ctypes.c_float, _vp, _vp, _i64, _vp, _vp, _vp, _i64, _vp]),
    "lvs_sort_keys_desc": (_i32, [_vp, _i64, _i32, _vp]),
    "lvs_certify_topk": (_i32, [_vp, _vp, _vp, _i64, _i32, _i32, ctypes.c_float, ctypes.c_float, _vp, _vp, _vp]),
    "lvs_certify_topk_banded": (_i32, [_vp, _vp, _vp, _i64, _i32, _i32, ctypes.c_float, ctypes.c_float, ctypes.c_float, _vp,
                                       _vp, _vp]),
    "lvs_margin_select": (_i32, [_vp, _vp, _vp, _i64, ctypes.c_float, ctypes.c_float, _vp, _vp, _vp]),
    "lvs_margin_select_stats": (_i32, [_vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp]),
    "lvs_kmeans_accumulate_workspace_bytes": (_i64, [_i64, _i32]),
    "lvs_kmeans_accumulate": (_i32, [_vp, _i64, _i32, _i32, _vp, _i32, _vp, _vp, _vp, _i64, _vp]),
    "lvs_kmeans_accumulate_keys": (_i32, [_vp, _i64, _i32, _i32, _vp, _i64, _i32, _vp, _vp, _vp, _i64, _vp]),
    "lvs_kmeans_objective_workspace_bytes": (_i64, [_i32]),
    "lvs_kmeans_objective": (_i32, [_vp, _vp, _vp, _i32, _i32, _vp, _vp, _vp, _i64, _vp]),
    "lvs_kmeans_pack_centroids": (_i32, [_vp, _i32, _i32, _i32, _vp, _vp, _vp, _vp]),
    "lvs_kmeans_update_centroids": (_i32, [_vp, _vp, _i32, _i32, _i64, _vp, _vp, _i32, _vp, _vp, _vp, _vp]),
    "lvs_kmeans_centroid_shift": (_i32, [_vp, _vp, _i32, _i32, _vp, _vp, _vp]),
    "lvs_kmeans_bounds_set": (_i32, [_vp, _i32, _vp, _vp, _vp, _i64, _vp, _vp, _i64, _vp, _vp, _vp, _vp]),
    "lvs_kmeans_bounds_fix": (_i32, [_vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp]),
    "lvs_kmeans_bounds_step": (_i32, [_vp, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp]),
    "lvs_rand_perm_host": (_i32, [_i64, _i64, _vp]),
    "lvs_rand_perm_prefix_host": (_i32, [_i64, _i64, _i64, _vp]),
    "lvs_kmeans_split_clusters_host": (_i32, [_i32, _i32, _i64, _vp, _vp, ctypes.POINTER(_i32)]),
    "lvs_kmeans_iteration_workspace_bytes": (_i64, [_i64, _i32, _i32, _i32, _i32]),
    "lvs_kmeans_iteration": (_i32, [_vp, _vp, _vp, _i32, _i64, _i32, _vp, _vp, _i32, _i32, _i64, _vp, _i32, _vp, _vp, _vp, _vp, _vp,
                                    _vp, _vp, _vp, _i64, _vp]),
    "lvs_kmeans_iteration_rccl": (_i32, [_vp, _vp, _i32, _i64, _i32, _vp, _vp, _i32, _i32, _i64, _vp, _i32, _vp, _vp, _vp, _vp, _vp,
                                         _vp, _vp, _vp, _i64, _vp]),
    "lvs_rccl_bind_all_reduce": (_i32, [_vp]),
    "lvs_timing_enable": (_i32, [_i32]),
    "lvs_timing_read": (_i32, [ctypes.POINTER(_dbl), ctypes.POINTER(_i64)]),
    "lvs_timing_read_calls": (_i32, [ctypes.POINTER(_dbl), ctypes.POINTER(_i64), ctypes.POINTER(_i64), ctypes.POINTER(_i32)]),
}


class LotusHipError(RuntimeError):
    """A C-ABI call returned a non-zero status."""


_lib = None


def declared_symbols() -> list[str]:
    """Function names declared in include/lotus_hip.h."""
    with open(HEADER_PATH) as f:
        text = f.read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(lvs_[a-z0-9_]+)\s*\(", text)))


def load(path: str | None = None):
    """dlopen the library and bind every declared symbol.  Raises if the library is missing.

    ``path`` selects another build explicitly (development: the ``make tuning`` library with the LVS_* knobs); the
    default library must be the shipped build - one that reports tuning knobs is refused."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    explicit = path is not None
    path = path or LIB_PATH
    if not os.path.exists(path):
        raise LotusHipError(
            f"{path} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). lotus_amd has no CPU fallback.")
    # PyTorch-ROCm wheels bundle their own libamdhip64.so.  The device buffers and streams this library receives come
    # from torch, so both must share ONE HIP runtime: load torch's first, then our NEEDED libamdhip64.so.7 resolves to
    # the copy already in the process.  (Loaded the other way round, our kernels run on a second runtime that knows
    # nothing about torch's context: "no ROCm-capable device is detected".)
    try:
        import torch  # noqa: F401
    except Exception:  # symbol checks still work without torch
        pass
    lib = ctypes.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    if lib.lvs_abi_version() != ABI_VERSION:
        raise LotusHipError(f"ABI version mismatch: library reports {lib.lvs_abi_version()}, binding is {ABI_VERSION}")
    if (lib.lvs_build_flags() & BUILD_TUNING) and not explicit:
        raise LotusHipError(f"{path} is a tuning build (environment knobs that can corrupt results); rebuild with `make`")
    _lib = lib
    return lib


def check(status: int, what: str = "") -> None:
    if status != OK:
        msg = load().lvs_last_error().decode("utf-8", "replace")
        raise LotusHipError(f"{what or 'lotus_hip call'} failed with status {status}: {msg}")


def device_count() -> int:
    n = _i32(0)
    st = load().lvs_device_count(ctypes.byref(n))
    return int(n.value) if st == OK else 0
