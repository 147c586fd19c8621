"""Five full-data k-means iterations at BASELINE configs[4]'s size (10 M x 768 fp16 points, K = 1 024, fp32-accurate
centroids) for `rocprofv3 --kernel-trace --stats` (development aid): which kernels an iteration consists of.
usage: rocprofv3 --kernel-trace --stats --output-format csv -d OUT -o km -- python tools/kmeans_iter_workload.py [rows]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from lotus_amd import _capi
from lotus_amd.backend import HipBackend
from lotus_amd.cluster import kmeans

be = HipBackend("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
if "blobs" in sys.argv[1:]:  # the bench's configs[4] rows (host-generated numpy streams)
    import benchdata
    xh, _ = benchdata.blobs(benchdata.CFG_KMEANS, n, 768, 1024)
    pk = be.pack(xh, _capi.PACK_F16)
    del xh
else:
    g = torch.Generator(device=be.device); g.manual_seed(7)
    x = torch.nn.functional.normalize(torch.randn((n, 768), generator=g, device=be.device), dim=1).to(torch.float16)
    pk = be.pack(x, _capi.PACK_F16)
    del x
kw = dict(backend=be, packed=pk, max_points_per_centroid=None, final_assign=False, bounds=False)
kmeans(None, 1024, niter=1, **kw); be.synchronize()
if "bounds" in sys.argv[1:]:  # for rocprofv3: one bounded run of 12 iterations
    t0 = time.perf_counter(); st = {}
    kmeans(None, 1024, niter=12, **dict(kw, bounds=True, stats=st)); be.synchronize()
    print(f"12 iterations with bounds: {(time.perf_counter() - t0) * 1e3:.1f} ms searched {[round(v / n, 3) for v in st['searched_rows']]} "
          f"uncertified {st['uncertified']} of {st['queries']}", flush=True)
    sys.exit(0)
t0 = time.perf_counter(); kmeans(None, 1024, niter=5, **kw); be.synchronize()
print(f"5 iterations (exhaustive): {(time.perf_counter() - t0) * 1e3:.1f} ms", flush=True)
if "both" in sys.argv[1:]:
    for b in (False, True):
        for niter in (2, 6, 2, 6, 12):
            st = {}
            be.synchronize(); t0 = time.perf_counter()
            kmeans(None, 1024, niter=niter, **dict(kw, bounds=b, stats=st)); be.synchronize()
            print(f"bounds={b} niter={niter}: {(time.perf_counter() - t0) * 1e3:.1f} ms searched {[round(v / n, 3) for v in st.get('searched_rows', [])]}", flush=True)
