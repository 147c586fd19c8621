"""fp32 embeddings (hi|lo rows), 10 k x 1 M x 768, k = 10: a few certified one-pass searches for `rocprofv3 --kernel-trace --stats`
(development aid): which launches the 2 ms above the fp16 search consist of."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lotus_amd import _capi
from lotus_amd.backend import HipBackend

be = HipBackend("cuda:0")
n, nq, d, k = 1_000_000, 10_000, 768, 10
g = torch.Generator(device=be.device); g.manual_seed(3)
xb = torch.nn.functional.normalize(torch.randn((n, d), generator=g, device=be.device), dim=1)
j = torch.randint(0, n, (nq,), generator=g, device=be.device)
xq = torch.nn.functional.normalize(0.7 * xb[j] + 0.7 * torch.nn.functional.normalize(torch.randn((nq, d), generator=g, device=be.device), dim=1), dim=1)
c32, q32 = be.pack(xb, _capi.PACK_SPLIT), be.pack(xq, _capi.PACK_SPLIT)
c16, q16 = be.pack(xb.half(), _capi.PACK_F16), be.pack(xq.half(), _capi.PACK_F16)
del xb, xq
for tag, c, q in (("fp16", c16, q16), ("fp32 one-pass", c32, q32)):
    for _ in range(2):
        be.search_keys(c, q, k, 0)
    be.synchronize()
    st = {}
    t0 = time.perf_counter()
    for _ in range(5):
        be.search_keys(c, q, k, 0, stats=st)
    be.synchronize()
    print(f"{tag}: {(time.perf_counter() - t0) / 5 * 1e3:.2f} ms per call  {st}", flush=True)
for spare in (1, 2, 3, 5):
    be.CERT_SPARE_SMALL_K = spare
    for _ in range(2):
        be.search_keys(c32, q32, k, 0)
    be.synchronize()
    st = {}
    t0 = time.perf_counter()
    for _ in range(5):
        be.search_keys(c32, q32, k, 0, stats=st)
    be.synchronize()
    print(f"fp32 one-pass, k1 = k + {spare}: {(time.perf_counter() - t0) / 5 * 1e3:.2f} ms per call  {st}", flush=True)
