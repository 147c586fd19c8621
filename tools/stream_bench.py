"""Small-batch (HBM-bound) search timing: nq queries x nb rows x 768 fp16, kernel time by HIP events (development aid)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lotus_amd.backend import HipBackend
from lotus_amd import _capi
be = HipBackend("cuda:0")
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
g = torch.Generator(device=be.device); g.manual_seed(1)
xb = torch.empty((nb, 768), dtype=torch.float16, device=be.device)
for r0 in range(0, nb, 1 << 18):
    r1 = min(nb, r0 + (1 << 18))
    xb[r0:r1] = torch.nn.functional.normalize(torch.randn((r1 - r0, 768), generator=g, device=be.device), dim=1).to(torch.float16)
cb = be.pack(xb, _capi.PACK_F16)
for nq in (1, 8, 32):
    cq = be.pack(xb[:nq].clone(), _capi.PACK_F16)
    for _ in range(3): be.search_keys(cb, cq, 10, 0)
    be.synchronize(); ts = []
    for _ in range(8):
        be.timing_enable(True); be.search_keys(cb, cq, 10, 0); be.synchronize()
        tot, cnt = be.timing_read(); ts.append(tot / max(cnt, 1))
    be.timing_enable(False)
    t = min(ts)
    print(f"nq={nq:3d} nb={nb}: kernel {t*1e3:8.1f} us  {nb*768*2/(t*1e-3)/1e12:6.2f} TB/s", flush=True)
