"""Timing ablations of lvs_rq_kernel (tuning build; results of the ablated runs are wrong on purpose)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from lotus_amd import _capi
_capi.load(os.path.join(ROOT, "lotus_amd", "liblotus_hip_tuning.so"))
from lotus_amd.backend import HipBackend
be = HipBackend("cuda:0")
g = torch.Generator(device=be.device); g.manual_seed(5)
def unit(n, d):
    out = torch.empty((n, d), dtype=torch.float16, device=be.device)
    for r0 in range(0, n, 1 << 18):
        r1 = min(n, r0 + (1 << 18))
        out[r0:r1] = torch.nn.functional.normalize(torch.randn((r1 - r0, d), generator=g, device=be.device), dim=1).half()
    return out
cb = be.pack(unit(1_000_000, 768), _capi.PACK_F16)
xq = unit(256, 768)
os.environ["LVS_RQ"] = "1"
for rnd in range(1):
  for nq in (128, 256):
    cq = be.pack(xq[:nq].contiguous(), _capi.PACK_F16)
    for dbg, ad4, what in (("0", "0", "full"), ("0", "3", "A ring of 3 (four waves)"), ("5", "0", "no side words"), ("5", "3", "no side words, A ring 3"),
                           ("4", "0", "no epilogue"), ("3", "0", "no staging loads")):
        os.environ["LVS_RQ_DEBUG"] = dbg
        os.environ["LVS_RQ_AD4"] = ad4
        be.search_keys(cb, cq, 10, 0); be.synchronize()
        be.timing_enable(True)
        for _ in range(20):
            be.search_keys(cb, cq, 10, 0)
        be.synchronize()
        tot, cnt = be.timing_read(); be.timing_enable(False)
        print(f"{nq} queries, {what:28s}: kernel {tot / max(cnt, 1):6.3f} ms", flush=True)
os.environ["LVS_RQ_DEBUG"] = "0"
os.environ["LVS_RQ_AD4"] = "0"
os.environ["LVS_RQ_STAMPS"] = "1"
for mode in ("0", "2"):
    os.environ["LVS_RQ_MODE"] = mode
    for nq in (128, 256):
        cq = be.pack(xq[:nq].contiguous(), _capi.PACK_F16)
        for _ in range(2):
            be.search_keys(cb, cq, 10, 0); be.synchronize()
os.environ["LVS_RQ_STAMPS"] = "0"
