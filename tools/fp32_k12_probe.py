"""fp32 embeddings, 10 k x 1 M x 768, k = 11 / 12 / 13: 15-slot banded lists on the 256-query geometry against k + 8 slots on the
128-query geometry (development aid, same box)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lotus_amd import _capi
from lotus_amd.backend import HipBackend

be = HipBackend("cuda:0")
n, nq, d = 1_000_000, 10_000, 768
g = torch.Generator(device=be.device); g.manual_seed(3)
xb = torch.nn.functional.normalize(torch.randn((n, d), generator=g, device=be.device), dim=1)
j = torch.randint(0, n, (nq,), generator=g, device=be.device)
xq = torch.nn.functional.normalize(0.7 * xb[j] + 0.7 * torch.nn.functional.normalize(torch.randn((nq, d), generator=g, device=be.device), dim=1), dim=1)
c32, q32 = be.pack(xb, _capi.PACK_SPLIT), be.pack(xq, _capi.PACK_SPLIT)
del xb, xq
for k in (10, 11, 12, 13):
    ref = be.search_keys(c32, q32, k, 0, one_pass=False)
    for rule, lim in (("15 slots", 13), ("k + 8 slots", 10)):  # CERT_MAX_K_SMALL_LISTS forced either way
        be.CERT_MAX_K_SMALL_LISTS = lim
        for _ in range(2):
            out = be.search_keys(c32, q32, k, 0)
        be.synchronize()
        st = {}
        t0 = time.perf_counter()
        for _ in range(5):
            out = be.search_keys(c32, q32, k, 0, stats=st)
        be.synchronize()
        same = float((out == ref).float().mean().item())
        print(f"k = {k}, {rule}: {(time.perf_counter() - t0) / 5 * 1e3:.2f} ms per call  {st}  keys equal to the plain search {same:.5f}", flush=True)
