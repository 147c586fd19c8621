"""Round-4 knob sweeps on the TUNING build (development aid): per shape a list of environment settings, kernel ms (library
HIP events) and wall ms per call, results compared bit for bit with the first variant.
usage: python tools/r05_sweep.py [mid|small|gq ...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from lotus_amd import _capi
_capi.load(os.path.join(ROOT, "lotus_amd", "liblotus_hip_tuning.so"))
from lotus_amd.backend import HipBackend

be = HipBackend("cuda:0")
what = sys.argv[1:] or ["small", "mid", "gq"]
nb, d, k = 1_000_000, 768, 10
g = torch.Generator(device=be.device); g.manual_seed(1)
xb = torch.nn.functional.normalize(torch.randn((nb, d), generator=g, device=be.device), dim=1).to(torch.float16)
nmax = 100_000
j = torch.randint(0, nb, (nmax,), generator=g, device=be.device)
xq = torch.nn.functional.normalize(0.7 * xb[j].float() + 0.7 * torch.nn.functional.normalize(torch.randn((nmax, d), generator=g, device=be.device), dim=1), dim=1).to(torch.float16)
cb, cq = be.pack(xb, _capi.PACK_F16), be.pack(xq, _capi.PACK_F16)
del xb, xq
KNOBS = ("LVS_L2_MIN_FINAL", "LVS_L2_SHORT_NQT", "LVS_LEAD", "LVS_GQ", "LVS_NSLAB", "LVS_L2_MIN_SLABS", "LVS_L2_MIN_TILES", "LVS_TAIL", "LVS_PLAN_PRINT")


def run(q, reps):
    for _ in range(2):
        be.keys_to_result(be.search_keys(cb, q, k, 0), 0)
    be.synchronize()
    best_k, best_w, keys = 1e9, 1e9, None
    for _ in range(3):
        be.timing_enable(True)
        t0 = time.perf_counter()
        for _ in range(reps):
            keys = be.search_keys(cb, q, k, 0)
            be.keys_to_result(keys, 0)
        be.synchronize()
        w = (time.perf_counter() - t0) / reps * 1e3
        tot, cnt = be.timing_read(); be.timing_enable(False)
        best_k, best_w = min(best_k, tot / max(cnt, 1)), min(best_w, w)
    return best_k, best_w, keys


def sweep(nq, variants, reps, rows=None):
    global cb
    full = cb
    if rows is not None:
        cb = be.slice_rows(full, 0, rows)
    try:
        _sweep(nq, variants, reps)
    finally:
        cb = full


def _sweep(nq, variants, reps):
    q = be.slice_rows(cq, 0, nq)
    ref = None
    for tag, env in variants:
        for kk in KNOBS:
            os.environ.pop(kk, None)
        os.environ.update(env)
        os.environ["LVS_PLAN_PRINT"] = "1"
        be.search_keys(cb, q, k, 0)  # prints the plan once
        os.environ.pop("LVS_PLAN_PRINT")
        km, wm, keys = run(q, reps)
        if ref is None:
            ref = keys.clone()
        fl = 2.0 * nq * cb.n * d
        print(f"{nq:>6} x {cb.n:>7}  {tag:<34} kernel {km:8.3f} ms  wall {wm:8.3f} ms  {fl / (km * 1e-3) / 1e12:7.1f} TFLOP/s  "
              f"{nb * d * 2 / (km * 1e-3) / 1e12:5.2f} TB/s  identical={bool((keys == ref).all())}", flush=True)


if "small" in what:
    # (round 4: a ring of three staging buffers on the 128-query geometry - two K-steps of loads in flight - and 128-query tiles
    # beyond 128 queries were built and measured here: no gain at 128 queries, 25-45 % slower beyond; profiles/r05c_sweep.log)
    for nq in (128, 192, 256, 384, 512, 1024, 2048):
        sweep(nq, [("shipped", {})], 20 if nq <= 512 else 10)
if "mid" in what:
    for nq in (20_000, 25_000, 30_000, 50_000):
        v = [("shipped", {}), ("no lead slab", {"LVS_LEAD": "0"}), ("wide groups (no L2 path)", {"LVS_L2_MIN_SLABS": "100000"}),
             ("L2 path, slabs >= 80 tiles", {"LVS_L2_MIN_TILES": "80"}), ("L2 path, slabs >= 160 tiles", {"LVS_L2_MIN_TILES": "160"}),
             ("gq 4", {"LVS_GQ": "4"}), ("gq 16", {"LVS_GQ": "16"})]
        sweep(nq, v, 3)
if "gq" in what:
    v = [("shipped (8 x 4 groups, lead slab)", {}), ("4 x 8 groups", {"LVS_GQ": "4"}), ("16 x 2 groups", {"LVS_GQ": "16"}),
         ("32 x 1 groups", {"LVS_GQ": "32"}), ("8 x 4, 9 slabs", {"LVS_NSLAB": "9"}), ("8 x 4, 17 slabs", {"LVS_NSLAB": "17"}),
         ("8 x 4, 25 slabs", {"LVS_NSLAB": "25"})]
    sweep(100_000, v, 2)

if "plan" in what:
    # (round 4 also swept a larger per-item cost in the slab-count model - fewer, longer slabs: slower everywhere,
    # profiles/r05d_plan_sweep.log)
    V = [("shipped", {}), ("8 x 4 groups whatever the slab length", {"LVS_L2_MIN_FINAL": "0"}),
         ("wide groups under 120-tile slabs at any size", {"LVS_L2_SHORT_NQT": "100000"})]
    for nq, rows in ((100_000, 125_000), (100_000, 250_000), (50_000, 250_000), (25_000, 500_000), (12_500, None), (20_000, None),
                     (25_000, None), (10_000, None), (4096, None), (100_000, None)):
        sweep(nq, V, 2 if nq * (rows or nb) >= 2e10 else 4, rows)
