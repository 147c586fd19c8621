"""Ten searches of NQ queries x 1 M rows x 768 fp16, k = 10, for a rocprofv3 --kernel-trace --stats run (development aid).
usage: python tools/midbatch_trace.py NQ"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from lotus_amd import _capi
from lotus_amd.backend import HipBackend
be = HipBackend("cuda:0")
nq, nb, d, k = int(sys.argv[1]), 1_000_000, 768, 10
g = torch.Generator(device=be.device); g.manual_seed(1)
xb = torch.nn.functional.normalize(torch.randn((nb, d), generator=g, device=be.device), dim=1).to(torch.float16)
j = torch.randint(0, nb, (nq,), generator=g, device=be.device)
xq = torch.nn.functional.normalize(0.7 * xb[j].float() + 0.7 * torch.nn.functional.normalize(torch.randn((nq, d), generator=g, device=be.device), dim=1), dim=1).to(torch.float16)
cb, cq = be.pack(xb, _capi.PACK_F16), be.pack(xq, _capi.PACK_F16)
for _ in range(3):
    be.search_keys(cb, cq, k, 0)
be.synchronize()
be.timing_enable(True)
t0 = time.perf_counter()
for _ in range(10):
    be.search_keys(cb, cq, k, 0)
be.synchronize()
wall = (time.perf_counter() - t0) / 10 * 1e3
f = be.timing_read_full(); be.timing_enable(False)
print(f"nq {nq}: wall {wall:.3f} ms per call, dominant kernel {f}", flush=True)
