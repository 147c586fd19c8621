"""Kernel time of `search_keys` at a list of shapes with a given build of the library (development aid for A/B runs:
call it alternately with two library paths on the same box).
usage: python tools/ab_libs.py LIB.so QxN [QxN ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from lotus_amd import _capi
_capi.load(os.path.abspath(sys.argv[1]))
from lotus_amd.backend import HipBackend

be = HipBackend("cuda:0")
d, k = 768, int(os.environ.get("AB_K", "10"))
METRIC = int(os.environ.get("AB_METRIC", "0"))  # 0 inner product, 1 L2
for shape in sys.argv[2:]:
    nq, nb = (int(v) for v in shape.split("x"))
    g = torch.Generator(device=be.device); g.manual_seed(1)
    xb = torch.nn.functional.normalize(torch.randn((nb, d), generator=g, device=be.device), dim=1).to(torch.float16)
    j = torch.randint(0, nb, (nq,), generator=g, device=be.device)
    u = torch.nn.functional.normalize(torch.randn((nq, d), generator=g, device=be.device), dim=1)
    xq = torch.nn.functional.normalize(0.7 * xb[j].float() + 0.7 * u, dim=1).to(torch.float16)
    cb, cq = be.pack(xb, _capi.PACK_F16), be.pack(xq, _capi.PACK_F16)
    del xb, xq, u
    for _ in range(2):
        be.search_keys(cb, cq, k, METRIC)
    be.synchronize()
    ts = []
    for _ in range(5):
        be.timing_enable(True)
        be.search_keys(cb, cq, k, METRIC)
        be.synchronize()
        tot, cnt = be.timing_read()
        ts.append(tot / max(cnt, 1))
    be.timing_enable(False)
    fl = 2.0 * nq * nb * d
    print(f"{os.path.basename(sys.argv[1]):28s} {shape:18s} min {min(ts):8.2f} ms  med {sorted(ts)[2]:8.2f} ms  "
          f"{fl / (min(ts) * 1e-3) / 1e12:7.1f} TFLOP/s", flush=True)
    del cb, cq
