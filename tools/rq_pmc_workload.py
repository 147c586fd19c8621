"""Four launches each of lvs_rq_kernel at RQ_PMC_NQ (default 128,256) queries x 1 M x 768 for rocprofv3 --pmc passes (shipped build)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from lotus_amd import _capi
from lotus_amd.backend import HipBackend
be = HipBackend("cuda:0")
g = torch.Generator(device=be.device); g.manual_seed(5)
def unit(n, d):
    out = torch.empty((n, d), dtype=torch.float16, device=be.device)
    for r0 in range(0, n, 1 << 18):
        r1 = min(n, r0 + (1 << 18))
        out[r0:r1] = torch.nn.functional.normalize(torch.randn((r1 - r0, d), generator=g, device=be.device), dim=1).half()
    return out
cb = be.pack(unit(1_000_000, 768), _capi.PACK_F16)
sizes = [int(v) for v in os.environ.get("RQ_PMC_NQ", "128,256").split(",")]
xq = unit(max(sizes), 768)
for nq in sizes:
    cq = be.pack(xq[:nq].contiguous(), _capi.PACK_F16)
    for _ in range(4):
        be.search_keys(cb, cq, 10, 0)
    be.synchronize()
