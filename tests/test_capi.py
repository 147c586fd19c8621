"""The C-ABI library builds for gfx950 without a GPU, loads, and exports exactly what include/lotus_hip.h declares.
No compute call is made here."""
import ctypes
import os
import subprocess

from lotus_amd import _capi


def test_library_exports_every_declared_symbol():
    lib = _capi.load()
    declared = _capi.declared_symbols()
    assert len(declared) >= 14
    assert set(declared) == set(_capi.SIGNATURES), "ctypes table and header out of sync"
    for name in declared:
        assert hasattr(lib, name), name


def test_shared_object_is_gfx950_code():
    out = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "--notes", _capi.LIB_PATH], capture_output=True, text=True)
    if out.returncode != 0:  # tool missing: fall back to a byte search
        blob = open(_capi.LIB_PATH, "rb").read()
        assert b"gfx950" in blob
    else:
        blob = open(_capi.LIB_PATH, "rb").read()
        assert b"gfx950" in blob and b"lvs_tile_kernel" in blob


def test_host_side_entry_points_without_a_gpu():
    lib = _capi.load()
    assert lib.lvs_abi_version() == _capi.ABI_VERSION == 7 and lib.lvs_build_flags() == 0
    assert lib.lvs_packed_ld(768, _capi.PACK_F16) == 768
    assert lib.lvs_packed_ld(100, _capi.PACK_F16) == 128
    assert lib.lvs_packed_ld(384, _capi.PACK_SPLIT) == 768
    assert lib.lvs_packed_ld(0, _capi.PACK_F16) < 0 and lib.lvs_packed_ld(8, 7) < 0
    ws = lib.lvs_flat_search_workspace_bytes(100000, 1000000, 768, 10, 0, 0)
    assert 100000 * 10 * 8 <= ws < 1 << 30
    assert lib.lvs_flat_search_workspace_bytes(-1, 10, 8, 1, 0, 0) < 0
    assert lib.lvs_flat_search_workspace_bytes(10, 10, 8, 1, 5, 0) < 0  # bad pack mode
    # argument validation happens before any device work
    st = lib.lvs_flat_search_keys(None, 0, 10, None, 0, 10, 8, 5, 3, None, None, 0, None, None, None, 0, None)
    assert st == _capi.EINVAL and b"metric" in lib.lvs_last_error()
    st = lib.lvs_merge_keys(None, 2, 5, 100, None, None)
    assert st == _capi.EINVAL
    assert lib.lvs_flat_search_keys(None, 0, 10, None, 0, 0, 8, 0, 3, None, None, 0, None, None, None, 0, None) == 0
    # the sharded search: scratch = both all-gather images + this shard's lists + the search's own scratch; pooled thresholds
    # only for fp16 rows, k <= 56 and more than one rank (decided from the arguments, identically on every rank)
    one = lib.lvs_search_sharded_workspace_bytes(1, 100000, 1000000, 768, 10, 0, 0, 20)
    assert ws <= one < ws + 2 * 100000 * 10 * 8 + 4096
    w8 = lib.lvs_search_sharded_workspace_bytes(8, 100000, 125000, 768, 10, 0, 0, 20)
    w8_unseeded = lib.lvs_search_sharded_workspace_bytes(8, 100000, 125000, 768, 10, 0, 0, 0)
    assert w8 - w8_unseeded >= 9 * 20 * 100000 * 4 and w8_unseeded >= 9 * 100000 * 10 * 8
    assert lib.lvs_search_sharded_workspace_bytes(8, 100000, 125000, 768, 10, _capi.PACK_SPLIT, 0, 20) < w8  # hi|lo rows: no pool
    assert lib.lvs_search_sharded_workspace_bytes(0, 10, 10, 8, 1, 0, 0, 0) < 0
    st = lib.lvs_search_sharded(None, None, 2, None, 0, 10, None, 0, 10, 8, 0, 3, None, None, 0, 0, None, None, 0, None)
    assert st == _capi.EINVAL and b"all-gather" in lib.lvs_last_error()
    assert lib.lvs_search_sharded(None, None, 1, None, 0, 10, None, 0, 0, 8, 0, 3, None, None, 0, 0, None, None, 0, None) == 0
    assert lib.lvs_rccl_available() in (0, 1)
    assert lib.lvs_search_sharded_rccl(None, None, 0, 10, None, 0, 10, 8, 0, 3, None, None, 0, 0, None, None, 0, None) != 0


def test_missing_library_fails_loudly(monkeypatch):
    import pytest

    monkeypatch.setattr(_capi, "_lib", None)
    monkeypatch.setattr(_capi, "LIB_PATH", "/nonexistent/liblotus_hip.so")
    with pytest.raises(_capi.LotusHipError):
        _capi.load()


def test_no_product_module_imports_the_oracle():
    root = os.path.dirname(os.path.abspath(_capi.__file__))
    for dirpath, _, files in os.walk(root):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                text = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in text and "from oracle" not in text, f


def test_workspace_sizes_cover_every_routing_decision():
    """`lvs_flat_search_workspace_bytes` is pure host arithmetic: it must accept every shape the search accepts and be
    large enough for the path the search will take (k <= 15, 16..56, and the two-phase path beyond 56, whose lists and
    buckets dominate)."""
    lib = _capi.load()
    f = lambda nq, nb, d, k, bm=0, qm=0: lib.lvs_flat_search_workspace_bytes(nq, nb, d, k, bm, qm)  # noqa: E731
    base = f(100_000, 1_000_000, 768, 10)
    assert base > 100_000 * 4 + 11 * 100_000 * 10 * 8 // 2  # thresholds + per-slab candidate lists
    assert f(100_000, 1_000_000, 768, 24) > 0 and f(100_000, 1_000_000, 768, 56) > 0
    big = f(100_000, 1_000_000, 768, 100)
    # two-phase: >= 2k/15 slabs x 15 keys per query of lists + a 512-slot bucket per query, 8 B each
    assert big >= 100_000 * (14 * 15 + 512) * 8
    assert f(100_000, 1_000_000, 768, 2048) >= 100_000 * 4096 * 8
    assert f(1, 1_000_000, 768, 1000) < 1 << 30          # one query: small
    assert f(0, 1000, 64, 5) >= 0 and f(10, 0, 64, 5) >= 0  # empty sides are legal
    assert f(10, 1000, 0, 5) < 0 and f(10, 1000, 64, -1) < 0  # bad shapes are refused
    # beyond 24 GB the two-phase path is not planned: the call falls back to selection passes and their small workspace
    assert f(4_000_000, 1_000_000, 768, 2048) < 24 << 30
    # pack modes are part of the contract (ABI 2); sizes must never shrink below the fp16 case
    for k in (1, 10, 56, 100):
        assert f(5000, 300_000, 384, k, 1, 1) >= f(5000, 300_000, 384, k, 0, 0) > 0


KNOBS = ["LVS_DEBUG_HOT", "LVS_STREAM_DEBUG", "LVS_GQ", "LVS_NSLAB", "LVS_TOP1", "LVS_TWO_PHASE", "LVS_SMALLQ",
         "LVS_LEAD", "LVS_L2_MIN_TILES", "LVS_STREAM", "LVS_STREAM_WGS", "LVS_COUNT", "LVS_TWO_PHASE_SLOTS"]


def test_shipped_library_has_no_tuning_or_debug_knob():
    """one stray LVS_* environment variable must not be able to change (or corrupt) results.
    The shipped .so contains none of the knob names and reports build flags 0; the names only exist in the separate
    `make tuning` build."""
    blob = open(_capi.LIB_PATH, "rb").read()
    for name in KNOBS:
        assert (name.encode() + b"\0") not in blob, name
    assert _capi.load().lvs_build_flags() == 0
    # the only getenv left is rocPRIM's own (ROCPRIM_USE_ATOMIC_BLOCK_ID, header-inlined into the two sort users)
    out = subprocess.run(["nm", "-D", "--undefined-only", _capi.LIB_PATH], capture_output=True, text=True).stdout
    assert "atoll" not in out and "atoi" not in out


def test_tuning_build_is_refused_by_default(tmp_path):
    import pytest

    csrc = os.path.join(os.path.dirname(_capi.LIB_PATH), "csrc")
    r = subprocess.run(["make", "-C", csrc, "tuning", "-j4"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    tuning = os.path.join(os.path.dirname(_capi.LIB_PATH), "liblotus_hip_tuning.so")
    blob = open(tuning, "rb").read()
    assert b"LVS_DEBUG_HOT\0" in blob
    lib = ctypes.CDLL(tuning)
    lib.lvs_build_flags.restype = ctypes.c_int32
    assert lib.lvs_build_flags() & _capi.BUILD_TUNING
    saved, saved_path = _capi._lib, _capi.LIB_PATH
    try:
        _capi._lib, _capi.LIB_PATH = None, tuning
        with pytest.raises(_capi.LotusHipError, match="tuning build"):
            _capi.load()
        assert _capi.load(tuning).lvs_build_flags() & _capi.BUILD_TUNING  # explicit path: allowed (tools/)
    finally:
        _capi._lib, _capi.LIB_PATH = saved, saved_path


def test_merge_keys_accepts_long_lists_from_many_parts():
    """8 shards x k > 512 used to be rejected.  Argument validation only (no GPU here)."""
    lib = _capi.load()
    assert lib.lvs_merge_keys(None, 8, 0, 1000, None, None) == 0          # nq == 0: nothing to do, accepted
    assert lib.lvs_merge_keys(None, 8, 5, 4000, None, None) == _capi.EINVAL  # NULL buffers / k beyond LVS_MAX_K


def test_block_to_item_deal_covers_every_item_once(tmp_path):
    """The tile kernel's blockIdx -> (query tile, slab) deal (lvs_tile.h): every item exactly once for 1 716 combinations of
    query tiles x slabs x XCD group shape, and the planner's per-XCD round count equals a direct count (host code only)."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = tmp_path / "mapping_check"
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    subprocess.run([hipcc, "-O1", "-std=c++17", "--offload-arch=gfx950", "-w",
                    os.path.join(root, "tests", "native", "mapping_check.cpp"), "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout
    assert out.startswith("ok "), out


def test_permutation_prefix_equals_the_full_permutation():
    """lvs_rand_perm_prefix_host (O(m)) against lvs_rand_perm_host and the oracle's restatement of faiss rand_perm."""
    import numpy as np

    import oracle

    lib = _capi.load()
    for n, m, seed in [(1, 1, 3), (2, 1, 0), (10, 10, 1234), (1000, 7, 1), (1000, 999, 2), (100003, 256, 1234),
                       (100003, 4096, 1235), (300000, 262144, 99), (5, 0, 1)]:
        full = np.empty(n, np.int64)
        assert lib.lvs_rand_perm_host(n, seed, full.ctypes.data) == 0
        pre = np.full(max(m, 1), -1, np.int64)
        assert lib.lvs_rand_perm_prefix_host(n, seed, m, pre.ctypes.data) == 0
        assert np.array_equal(pre[:m], full[:m]), (n, m, seed)
        if n <= 100003:
            assert np.array_equal(full, oracle.rand_perm(n, seed))
    assert lib.lvs_rand_perm_prefix_host(10, 1, 11, None) == _capi.EINVAL
