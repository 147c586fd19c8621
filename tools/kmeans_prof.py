"""k-means full-data iterations under rocprofv3 (development aid): 10M x 768 fp16, K = 1024."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from lotus_amd import _capi
from lotus_amd.backend import HipBackend
from lotus_amd.cluster import kmeans
be = HipBackend("cuda:0"); dev = be.device
n = int(os.environ.get("KM_N", "10000000"))
g = torch.Generator(device=dev); g.manual_seed(7)
x = torch.empty((n, 768), dtype=torch.float16, device=dev)
for r0 in range(0, n, 1 << 18):
    r1 = min(n, r0 + (1 << 18))
    x[r0:r1] = torch.nn.functional.normalize(torch.randn((r1 - r0, 768), generator=g, device=dev), dim=1).to(torch.float16)
pk = be.pack(x, _capi.PACK_F16)
xh = np.empty((n, 768), np.float16)
for r0 in range(0, n, 1 << 20): xh[r0:r0 + (1 << 20)] = x[r0:r0 + (1 << 20)].cpu().numpy()
del x
kw = dict(backend=be, packed=pk, pack_mode=_capi.PACK_F16, max_points_per_centroid=None, final_assign=False)
for prec in ("fp32", "fp16"):
    ts = {}
    for niter in (1, 5, 1, 5):
        be.synchronize(); t0 = time.perf_counter()
        kmeans(xh, 1024, niter=niter, centroid_precision=prec, **kw)
        be.synchronize(); ts[niter] = time.perf_counter() - t0
    print(prec, "centroids: setup + 1 iteration", ts[1], "s; per further iteration", (ts[5] - ts[1]) / 4, "s", flush=True)
