"""k-means timings with a given build of the library (development aid for A/B runs).
usage: python tools/ab_kmeans.py LIB.so [rows]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from lotus_amd import _capi
_capi.load(os.path.abspath(sys.argv[1]))
from lotus_amd.backend import HipBackend
from lotus_amd.cluster import kmeans

be = HipBackend("cuda:0")
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4_000_000
g = torch.Generator(device=be.device); g.manual_seed(7)
x = torch.nn.functional.normalize(torch.randn((n, 768), generator=g, device=be.device), dim=1).to(torch.float16)
pk = be.pack(x, _capi.PACK_F16)
del x
name = os.path.basename(sys.argv[1])
for rep in range(3):
    be.synchronize(); t0 = time.perf_counter()
    r = kmeans(None, 1024, niter=20, backend=be, packed=pk)
    be.synchronize(); t = time.perf_counter() - t0
    print(f"{name:26s} parity mode (262144-row subsample, 20 it + final assign of {n} rows): {t*1e3:8.1f} ms  obj {r.obj[-1]:.1f}", flush=True)
kw = dict(backend=be, packed=pk, max_points_per_centroid=None, final_assign=False)
ts = {}
for niter in (1, 5, 1, 5):
    be.synchronize(); t0 = time.perf_counter(); kmeans(None, 1024, niter=niter, **kw); be.synchronize(); ts[niter] = time.perf_counter() - t0
print(f"{name:26s} full-data per iteration: {(ts[5]-ts[1])/4*1e3:8.2f} ms", flush=True)
cent = be.pack(be.unpack(pk, be.to_device(np.arange(1024, dtype=np.int64))), _capi.PACK_SPLIT)
for exact in (True, False):
    be.nearest(cent, pk, _capi.METRIC_L2, exact_scores=exact); be.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        be.nearest(cent, pk, _capi.METRIC_L2, exact_scores=exact)
    be.synchronize()
    print(f"{name:26s} nearest({n} x 1024, exact_scores={exact}): {(time.perf_counter()-t0)/5*1e3:8.2f} ms", flush=True)
