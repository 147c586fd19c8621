"""Sweep the XCD group shape (LVS_GQ) and slab count (LVS_NSLAB) of the tile kernel at one shape (development aid)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lotus_amd.backend import HipBackend
from lotus_amd import _capi
be = HipBackend("cuda:0")
nq, nb = (int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "100000x1000000").split("x"))
d, k = 768, int(os.environ.get("QB_K", "10"))
g = torch.Generator(device=be.device); g.manual_seed(1)
xb = torch.nn.functional.normalize(torch.randn((nb, d), generator=g, device=be.device), dim=1).to(torch.float16)
j = torch.randint(0, nb, (nq,), generator=g, device=be.device)
xq = torch.nn.functional.normalize(0.7 * xb[j].float() + 0.7 * torch.nn.functional.normalize(torch.randn((nq, d), generator=g, device=be.device), dim=1), dim=1).to(torch.float16)
cb, cq = be.pack(xb, _capi.PACK_F16), be.pack(xq, _capi.PACK_F16)
ref = None
configs = [tuple(c.split(":")) for c in (sys.argv[2].split(",") if len(sys.argv) > 2 else ["32:", "16:", "8:", "4:", "8:24", "4:32", "32:"])]
for gq, ns in configs:
    os.environ["LVS_GQ"] = gq
    if ns: os.environ["LVS_NSLAB"] = ns
    else: os.environ.pop("LVS_NSLAB", None)
    for _ in range(2): keys = be.search_keys(cb, cq, k, 0)
    be.synchronize()
    ts = []
    for _ in range(4):
        be.timing_enable(True); keys = be.search_keys(cb, cq, k, 0); be.synchronize()
        tot, cnt = be.timing_read(); ts.append(tot / max(cnt, 1))
    be.timing_enable(False)
    if ref is None: ref = keys.clone()
    same = bool((keys == ref).all())
    t = min(ts)
    print(f"gq={gq:>2} nslab={ns or 'auto':>4}: {t:8.2f} ms  {2.0*nq*nb*d/(t*1e-3)/1e12:7.1f} TFLOP/s  identical={same}", flush=True)
