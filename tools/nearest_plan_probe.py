"""Development aid (tuning build): certified assignment 4 M points x 1 024 centroids under different (slab count, XCD group
shape) plans of the tile kernel - does sharing a point tile among the workgroups of an XCD cut the 4.1 x re-read?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from lotus_amd import _capi
_capi.load(os.path.join(ROOT, "lotus_amd", "liblotus_hip_tuning.so"))
from lotus_amd.backend import HipBackend
be = HipBackend("cuda:0")
g = torch.Generator(device=be.device); g.manual_seed(1)
n, K, d = 4_000_000, 1024, 768
x = torch.empty((n, d), dtype=torch.float16, device=be.device)
for r0 in range(0, n, 1 << 18):
    r1 = min(n, r0 + (1 << 18))
    x[r0:r1] = torch.nn.functional.normalize(torch.randn((r1 - r0, d), generator=g, device=be.device), dim=1).to(torch.float16)
pts = be.pack(x, _capi.PACK_F16)
cent, cst = be.kmeans_pack_centroids(torch.nn.functional.normalize(torch.randn((K, d), generator=g, device=be.device), dim=1), _capi.PACK_SPLIT)
for env in ({}, {"LVS_NSLAB": "2", "LVS_GQ": "16"}, {"LVS_NSLAB": "4", "LVS_GQ": "8"}, {"LVS_NSLAB": "4", "LVS_GQ": "4"}, {"LVS_NSLAB": "2", "LVS_GQ": "8"}, {"LVS_GQ": "8"}, {}):
    for kk in ("LVS_NSLAB", "LVS_GQ"):
        os.environ.pop(kk, None)
    os.environ.update(env)
    for _ in range(2): be.nearest(cent, pts, 1, exact_scores=False, corpus_stats=cst)
    be.synchronize(); be.timing_enable(True)
    for _ in range(3): be.nearest(cent, pts, 1, exact_scores=False, corpus_stats=cst)
    be.synchronize(); tot, cnt = be.timing_read(); be.timing_enable(False)
    print(f"{env}: TOP2 kernel {tot / max(cnt, 1):.3f} ms per launch ({cnt} timed launches)", flush=True)
