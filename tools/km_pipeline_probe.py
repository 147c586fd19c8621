"""Exhaustive k-means iteration, 10 M x 768 fp16 blob rows, K = 1 024 (development aid): the certificate tails of a row range
(count read-back, pair dot products, exact search of the open rows) on the main stream (round 4) against on the side stream
under the next range's search; alternated twice on one box, 8 iterations each timed by device events, the first dropped."""
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
ref = None
for rnd in range(2):
    for pipe in (False, True):
        cluster.PIPELINE_CERTIFICATES = pipe
        st = {"time_iterations": True}
        res = cluster.kmeans(None, K, niter=8, stats=st, **kw)
        be.synchronize()
        ms = st["iteration_ms"][1:]
        c = np.asarray(res.centroids)
        same = "" if ref is None else f"  centroids identical to the first run: {bool(np.array_equal(c, ref))}"
        if ref is None:
            ref = c.copy()
        print(f"certificate tails on the {'side' if pipe else 'main'} stream: median {np.median(ms):6.2f} ms  min {min(ms):6.2f}  max {max(ms):6.2f}"
              f"  pairs {st.get('pairs', 0) / max(1, st.get('queries', 1)):.4f} open {st.get('open', 0) / max(1, st.get('queries', 1)):.4f}{same}", flush=True)
