"""Same-box A/B of two builds of the library on the 100 k x 1 M join (kernel / call ms): python tools/ab_join.py <lib.so> [nq]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from lotus_amd import _capi
_capi.load(os.path.abspath(sys.argv[1]))
from lotus_amd.backend import HipBackend
be = HipBackend("cuda:0")
g = torch.Generator(device=be.device); g.manual_seed(20260930)
nq = int(sys.argv[2]) if len(sys.argv) > 2 else 100_000
nb, d = 1_000_000, 768
def unit(n):
    out = torch.empty((n, d), dtype=torch.float16, device=be.device)
    for r0 in range(0, n, 1 << 18):
        r1 = min(n, r0 + (1 << 18))
        out[r0:r1] = torch.nn.functional.normalize(torch.randn((r1 - r0, d), generator=g, device=be.device), dim=1).half()
    return out
xb = unit(nb); cb = be.pack(xb, _capi.PACK_F16)
j = torch.randint(0, nb, (nq,), generator=g, device=be.device)
xq = torch.empty((nq, d), dtype=torch.float16, device=be.device)
for r0 in range(0, nq, 1 << 16):
    r1 = min(nq, r0 + (1 << 16))
    u = torch.nn.functional.normalize(torch.randn((r1 - r0, d), generator=g, device=be.device), dim=1)
    xq[r0:r1] = torch.nn.functional.normalize(0.7 * xb[j[r0:r1]].float() + 0.7 * u, dim=1).half()
cq = be.pack(xq, _capi.PACK_F16)
for _ in range(3):
    be.search_keys(cb, cq, 10, 0)
be.synchronize(); be.timing_enable(True)
t0 = time.perf_counter(); n = 0
while time.perf_counter() - t0 < 3.0:
    be.search_keys(cb, cq, 10, 0); n += 1
be.synchronize(); wall = (time.perf_counter() - t0) / n * 1e3
tot, calls = be.timing_read()
print(f"{os.path.basename(sys.argv[1]):36s} kernel {tot / max(calls, 1):8.3f} ms  call {wall:8.3f} ms", flush=True)
