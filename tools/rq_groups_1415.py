"""14 and 15 query groups of lvs_rq_kernel (3 329 .. 3 840 queries) against the list kernel: keys bit for bit (tuning build)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from lotus_amd import _capi
_capi.load(os.path.join(ROOT, "lotus_amd", "liblotus_hip_tuning.so"))
from lotus_amd.backend import HipBackend
be = HipBackend("cuda:0")
g = torch.Generator(device=be.device); g.manual_seed(3)
def unit(n, d):
    return torch.nn.functional.normalize(torch.randn((n, d), generator=g, device=be.device), dim=1).half()
bad = 0
for d, nb in ((256, 600_001), (768, 500_000)):
    xb = unit(nb, d); xb[nb // 2] = xb[3]
    cb = be.pack(xb, _capi.PACK_F16)
    for nq in (3329, 3500, 3584, 3700, 3840):
        xq = unit(nq, d); xq[5] = xb[3]; xq[nq - 1] = xb[nb - 1]
        cq = be.pack(xq, _capi.PACK_F16)
        for metric in (0, 1):
            for k in (10, 16):
                os.environ["LVS_RQ"] = "0"; a = be.search_keys(cb, cq, k, metric, id_offset=7)
                os.environ["LVS_RQ"] = "1"; b = be.search_keys(cb, cq, k, metric, id_offset=7)
                be.synchronize()
                same = bool(torch.equal(a, b))
                bad += 0 if same else 1
                print(f"d={d} nq={nq} ({-(-nq // 256)} groups) metric={metric} k={k}: keys identical {same}", flush=True)
print("mismatching configurations:", bad)
