"""Development aid: step the bounded k-means loop by hand and compare every iteration's assignment with an exhaustive
search; print the first rows whose bounds claimed "unchanged" although the nearest centroid changed.
usage: python tools/kmeans_bounds_debug.py [rows] [k] [d]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import benchdata
from lotus_amd import _capi
from lotus_amd.backend import HipBackend

n = int(sys.argv[1]) if len(sys.argv) > 1 else 150_000
K = int(sys.argv[2]) if len(sys.argv) > 2 else 96
d = int(sys.argv[3]) if len(sys.argv) > 3 else 64
be = HipBackend("cuda:0"); dev = be.device
x, _ = benchdata.blobs(benchdata.CFG_KMEANS, n, d, K)
train = be.pack(x, _capi.PACK_F16)
perm = be.rand_perm(n, 1235, K)
centroids = be.unpack(train, be.to_device(perm[:K]), raw=True)
cmode = _capi.PACK_SPLIT
cpk, cstats = be.kmeans_pack_centroids(centroids, cmode)
b_assign = torch.full((n,), -1, dtype=torch.int32, device=dev)
b_ub = torch.zeros((n,), dtype=torch.float32, device=dev); b_lb = torch.zeros((n,), dtype=torch.float32, device=dev)
keys = None
x32 = torch.from_numpy(x.astype(np.float32)).to(dev)
for it in range(10):
    t0 = time.perf_counter()
    if keys is None:
        keys = be.nearest(cpk, train, 1, exact_scores=False, corpus_stats=cstats, bounds=(b_assign, b_ub, b_lb, None)); searched = n
    else:
        act = be.kmeans_bounds_step(b_assign, b_ub, b_lb, shift, top2)
        if act.numel() > n // 2:
            keys = be.nearest(cpk, train, 1, exact_scores=False, corpus_stats=cstats, bounds=(b_assign, b_ub, b_lb, None)); searched = n
        elif act.numel():
            sub = be.gather(train, act)
            keys[act] = be.nearest(cpk, sub, 1, exact_scores=False, corpus_stats=cstats, bounds=(b_assign, b_ub, b_lb, act)); searched = int(act.numel())
        else:
            searched = 0
    be.synchronize(); t1 = time.perf_counter()
    full = be.nearest(cpk, train, 1, exact_scores=False, corpus_stats=cstats)
    _, Ia = be.keys_to_result(keys, 1); _, Ib = be.keys_to_result(full, 1)
    bad = (Ia != Ib).reshape(-1).nonzero().reshape(-1)
    # true distances in float64 for the bad rows
    print(f"it {it}: searched {searched} ({searched / n:.3f}) in {(t1 - t0) * 1e3:.2f} ms; mismatches vs exhaustive: {bad.numel()}", flush=True)
    if bad.numel():
        c64 = centroids.double()
        for r in bad[:6].tolist():
            dist = ((x32[r].double()[None, :] - c64) ** 2).sum(1).sqrt()
            o = torch.argsort(dist)[:3]
            print(f"   row {r}: kept {int(Ia[r])} exhaustive {int(Ib[r])} b_assign {int(b_assign[r])} ub {float(b_ub[r]):.6f} lb {float(b_lb[r]):.6f}; true nearest {o.tolist()} at {dist[o].tolist()}"
                  f"; dist to kept {float(dist[int(Ia[r])]):.6f}")
    c_old = centroids.clone()
    sums, counts = be.kmeans_accumulate_keys(train, keys, K)
    ns = torch.zeros(1, dtype=torch.int32, device=dev)
    cpk, cstats = be.kmeans_finish(sums, counts, centroids, n, cmode, ns)
    shift, top2 = be.kmeans_centroid_shift(c_old, centroids)
    print(f"      splits {int(ns.item())}, max shift {float(top2[0]):.5f} (centroid {int(top2[1].view(torch.int32))}), second {float(top2[2]):.5f}", flush=True)
