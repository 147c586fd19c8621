import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from lotus_amd.backend import HipBackend
from lotus_amd import _capi
be = HipBackend("cuda:0"); dev = be.device
N, D = 4_000_000, 768
g = torch.Generator(device=dev); g.manual_seed(1)
x32 = torch.empty((N, D), dtype=torch.float32, device=dev)
for r0 in range(0, N, 1 << 18):
    x32[r0:r0 + (1 << 18)] = torch.nn.functional.normalize(torch.randn((min(1 << 18, N - r0), D), generator=g, device=dev), dim=1)
x16 = x32.to(torch.float16)
be.PACK_CHUNK_ROWS = N  # one launch per pack so that byte counts per launch are known
for _ in range(3):
    ps = be.pack(x32, _capi.PACK_SPLIT); del ps
    p16 = be.pack(x16, _capi.PACK_F16)
ids = torch.randperm(N, generator=g, device=dev)[:1_000_000].contiguous()
for _ in range(3): gth = be.gather(p16, ids)
assign = torch.randint(0, 1024, (N,), generator=g, device=dev)
for _ in range(3): be.kmeans_accumulate(p16, assign, 1024)
q = be.pack(x16[:1], _capi.PACK_F16)
for _ in range(5): keys = be.search_keys(p16, q, 10, 0)
k2 = torch.randint(1, 2**62, (100_000, 10), generator=g, device=dev)
for _ in range(3): be.keys_to_result(k2, 0)
be.synchronize()
