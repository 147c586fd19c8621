"""How to cut an exhaustive k-means iteration into row ranges (sums of one range under the search of the next; development
aid): 10 M x 768 fp16 blob rows, K = 1 024; per configuration 7 iterations timed by device events, the first dropped."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import benchdata
from lotus_amd import _capi, cluster
from lotus_amd.backend import HipBackend

be = HipBackend("cuda:0")
n, d, K = 10_000_000, 768, 1024
xh, _ = benchdata.blobs(benchdata.CFG_KMEANS, n, d, K)
pk = be.pack(xh, _capi.PACK_F16)
del xh
kw = dict(backend=be, packed=pk, max_points_per_centroid=None, final_assign=False, bounds=False)
cluster.kmeans(None, K, niter=2, **kw); be.synchronize()
lo, hi = torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else (0, -1)
print(f"stream priority range: least {lo}, greatest {hi}", flush=True)
ref = None
CONFIGS = [(4, 0), (1, 0), ((.3, .3, .25, .15), 0), ((.28, .28, .26, .18), 0), ((.34, .33, .23, .10), 0), (5, 0), ((.25, .25, .22, .18, .10), 0),
           (6, 0), (3, 0), (4, hi), (4, lo), ((.3, .3, .25, .15), lo), (4, 0)]
for parts, prio in CONFIGS:
    cluster.SIDE_STREAM_PRIORITY = prio
    st = {"time_iterations": True}
    try:
        res = cluster.kmeans(None, K, niter=7, parts=parts, stats=st, **kw)
    except Exception as e:  # e.g. a priority the runtime refuses
        print(f"parts {parts} priority {prio}: {type(e).__name__} {e}", flush=True)
        continue
    be.synchronize()
    ms = st["iteration_ms"][1:]
    c = res.centroids if not torch.is_tensor(res.centroids) else res.centroids.cpu().numpy()
    same = "" if ref is None else f"  centroids identical to the first run: {bool(np.array_equal(np.asarray(c), ref))}"
    if ref is None:
        ref = np.asarray(c).copy()
    print(f"parts {str(parts):32s} side priority {prio:2d}: median {np.median(ms):6.2f} ms  min {min(ms):6.2f}  max {max(ms):6.2f}{same}", flush=True)
