"""Seeded thresholds on BIG list launches (development aid, tuning build): 100 k queries x 125 k / 250 k / 1 M rows and
25 k x 500 k unseeded, as shipped, and with other sample sizes; kernel ms from the library's
HIP events, results compared bit for bit with the unseeded launch.  usage: python tools/seed_big_sweep.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from lotus_amd import _capi
_capi.load(os.path.join(ROOT, "lotus_amd", "liblotus_hip_tuning.so"))
from lotus_amd.backend import HipBackend

be = HipBackend("cuda:0")
d, k = 768, 10
g = torch.Generator(device=be.device); g.manual_seed(1)
nb, nq = 1_000_000, 100_000
xb = torch.nn.functional.normalize(torch.randn((nb, d), generator=g, device=be.device), dim=1).to(torch.float16)
j = torch.randint(0, nb, (nq,), generator=g, device=be.device)
xq = torch.nn.functional.normalize(0.7 * xb[j].float() + 0.7 * torch.nn.functional.normalize(torch.randn((nq, d), generator=g, device=be.device), dim=1), dim=1).to(torch.float16)
cb, cq = be.pack(xb, _capi.PACK_F16), be.pack(xq, _capi.PACK_F16)
del xb, xq
KNOBS = ("LVS_TILE_SEED", "LVS_TILE_SEED_DIV_BIG", "LVS_LEAD")
VARIANTS = (("unseeded", {"LVS_TILE_SEED": "0"}), ("shipped (max(k, 8) tiles up to 1.3 M rows)", {}),
            ("nb/128", {"LVS_TILE_SEED_DIV_BIG": "128"}), ("unseeded again", {"LVS_TILE_SEED": "0"}))

def run(c, q, reps):
    be.search_keys(c, q, k, 0); be.synchronize()
    be.timing_enable(True)
    t0 = time.perf_counter()
    for _ in range(reps):
        keys = be.search_keys(c, q, k, 0)
    be.synchronize()
    wall = (time.perf_counter() - t0) / reps * 1e3
    tot, cnt = be.timing_read(); be.timing_enable(False)
    return tot / max(cnt, 1), wall, keys


for rows, queries, reps in ((125_000, 100_000, 5), (250_000, 100_000, 3), (500_000, 25_000, 3), (1_000_000, 100_000, 2)):
    c, q = be.slice_rows(cb, 0, rows), be.slice_rows(cq, 0, queries)
    out, ref = [], None
    for tag, env in VARIANTS:
        for kk in KNOBS:
            os.environ.pop(kk, None)
        os.environ.update(env)
        kms, wms, keys = run(c, q, reps)
        same = "" if ref is None else (" =" if torch.equal(keys, ref) else " DIFFERENT")
        ref = keys if ref is None else ref
        out.append(f"{tag}: kernel {kms:8.3f} wall {wms:8.3f}{same}")
    print(f"{queries} x {rows}: " + "   ".join(out) + "   (ms)", flush=True)
