"""The assignment kernel's skewed folds (waves 0-3 fold at the start of the next K-step, beside the MFMAs of waves 4-7) against
folds of all eight waves at the tile's end; 10 M x 768 fp16 blob rows, K = 1 024; tuning build (LVS_ASSIGN_SKEW read per call).
(1) the kernel alone, from the library's own events; (2) the exhaustive iteration for a few range splits."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import benchdata
from lotus_amd import _capi
_capi.load(os.path.join(ROOT, "lotus_amd", "liblotus_hip_tuning.so"))
from lotus_amd import cluster
from lotus_amd.backend import HipBackend

be = HipBackend("cuda:0")
n, d, K = 10_000_000, 768, 1024
xh, _ = benchdata.blobs(benchdata.CFG_KMEANS, n, d, K)
pk = be.pack(xh, _capi.PACK_F16)
del xh
cent = be.unpack(pk, be.to_device(np.arange(K, dtype=np.int64)), raw=True)
cpk, cstats = be.kmeans_pack_centroids(cent, _capi.PACK_SPLIT)
ref = None
for rnd in range(2):
    for skew in ("0", "1"):
        os.environ["LVS_ASSIGN_SKEW"] = skew
        coef, dpad = be._nearest_coef(cpk, pk, _capi.METRIC_L2)
        one_pass = lambda: be._nearest3_begin(cpk, pk, _capi.METRIC_L2, 0, False, cstats, None, coef, dpad, {})["keys"]
        keys = one_pass()
        be.synchronize()
        be.timing_enable(True)
        for _ in range(4):
            keys = one_pass()
        be.synchronize()
        tot, cnt = be.timing_read()
        be.timing_enable(False)
        same = "" if ref is None else f"  keys identical: {bool(torch.equal(keys, ref))}"
        if ref is None:
            ref = keys.clone()
        print(f"skew {skew}: assignment kernel {tot / max(cnt, 1):7.3f} ms per launch ({cnt} launches){same}", flush=True)
kw = dict(backend=be, packed=pk, max_points_per_centroid=None, final_assign=False, bounds=False)
cluster.kmeans(None, K, niter=2, **kw); be.synchronize()
cref = None
for rnd in range(2):
    for parts in ((.3, .3, .25, .15), (.32, .32, .26, .10), (.27, .27, .26, .20), (.25, .25, .2, .18, .12)):
        for skew in ("0", "1"):
            os.environ["LVS_ASSIGN_SKEW"] = skew
            st = {"time_iterations": True}
            res = cluster.kmeans(None, K, niter=8, stats=st, parts=parts, **kw)
            be.synchronize()
            ms = st["iteration_ms"][1:]
            c = np.asarray(res.centroids)
            same = "" if cref is None else f"  centroids identical: {bool(np.array_equal(c, cref))}"
            if cref is None:
                cref = c.copy()
            print(f"parts {str(parts):28s} skew {skew}: median {np.median(ms):6.2f} ms  min {min(ms):6.2f}{same}", flush=True)
