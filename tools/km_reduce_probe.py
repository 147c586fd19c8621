"""km_reduce under different bucket-size distributions (development aid): 10 M x 768 fp16 rows, K = 1 024.
usage: python tools/km_reduce_probe.py [rows]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from lotus_amd import _capi
from lotus_amd.backend import HipBackend

be = HipBackend("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
K, d = 1024, 768
g = torch.Generator(device=be.device); g.manual_seed(3)
x = torch.randn((n, d), generator=g, device=be.device).to(torch.float16)
pk = be.pack(x, _capi.PACK_F16)
del x
rng = np.random.default_rng(5)


def sizes_to_assign(sz, shuffle=True):
    a = np.repeat(np.arange(K, dtype=np.int32), sz)
    if shuffle:
        rng.shuffle(a)
    return a


def timed(name, assign):
    t = torch.from_numpy(assign.astype(np.int64)).to(be.device)
    for _ in range(2):
        be.kmeans_accumulate(pk, t, K)
    be.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    for _ in range(5):
        be.kmeans_accumulate(pk, t, K)
    ev1.record(); be.synchronize()
    sz = np.bincount(assign, minlength=K)
    print(f"{name:44s} {ev0.elapsed_time(ev1) / 5:7.3f} ms per call   max bucket {sz.max() / (n / K):5.2f} x mean, empty {int((sz == 0).sum())}", flush=True)


m = n // K
uni = np.full(K, m); uni[: n - m * K] += 1
timed("uniform sizes, shuffled rows", sizes_to_assign(uni))
timed("uniform sizes, rows sorted by cluster", sizes_to_assign(uni, shuffle=False))
timed("random assignment (multinomial)", rng.integers(0, K, n).astype(np.int32))
one = uni.copy(); one[0] += 7 * m; one[1:] -= (7 * m) // (K - 1) + 1; one[1] += n - one.sum()
timed("uniform + ONE bucket 8 x", sizes_to_assign(one))
# the blob data's distribution after the first iterations: 13 % empty, the rest 1 x / 2 x / 3 x the base size
w = rng.choice([0, 1, 2, 3], size=K, p=[0.13, 0.62, 0.20, 0.05]).astype(np.float64)
blob = np.floor(w / w.sum() * n).astype(np.int64); blob[np.argmax(blob)] += n - blob.sum()
timed("blob-like sizes (13 % empty, 1 / 2 / 3 x)", sizes_to_assign(blob))
half = np.where(np.arange(K) % 2 == 0, 1.5, 0.5); hs = np.floor(half / half.sum() * n).astype(np.int64); hs[0] += n - hs.sum()
timed("alternating 1.5 x / 0.5 x", sizes_to_assign(hs))
