"""Exhaustive k-means iteration, 10 M x 768 fp16 blob rows, K = 1 024, with a given build of the library (development aid for
same-box A/B runs): 9 iterations timed one by one with device events, the first dropped.
usage: python tools/km_iter_ab.py LIB.so [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import benchdata
from lotus_amd import _capi
_capi.load(os.path.abspath(sys.argv[1]))
from lotus_amd import cluster
from lotus_amd.backend import HipBackend

be = HipBackend("cuda:0")
n, d, K = 10_000_000, 768, 1024
xh, _ = benchdata.blobs(benchdata.CFG_KMEANS, n, d, K)
pk = be.pack(xh, _capi.PACK_F16)
del xh
kw = dict(backend=be, packed=pk, max_points_per_centroid=None, final_assign=False, bounds=False)
for k, v in os.environ.items():
    if k.startswith("KM_") and hasattr(cluster, k[3:]):
        setattr(cluster, k[3:], eval(v)); print("cluster.%s = %s" % (k[3:], v))
cluster.kmeans(None, K, niter=2, **kw); be.synchronize()
for rep in range(int(sys.argv[2]) if len(sys.argv) > 2 else 2):
    st = {"time_iterations": True}
    res = cluster.kmeans(None, K, niter=9, stats=st, **kw)
    be.synchronize()
    ms = st["iteration_ms"][1:]
    print(f"{os.path.basename(sys.argv[1]):24s} iteration median {np.median(ms):6.2f} ms  min {min(ms):6.2f}  max {max(ms):6.2f}"
          f"  centroid checksum {float(np.asarray(res.centroids, dtype=np.float64).sum()):.9f}", flush=True)
