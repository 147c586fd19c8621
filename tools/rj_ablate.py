"""Timing ablations of lvs_rq_kernel at join scale (4 096 queries x 1 M x 768, 16 groups): the eight-wave form (LVS_RQ_MODE=0) and the
four-wave form with two query blocks per wave and the B fragments in named AGPRs (LVS_RQ_MODE=2).  Ablated runs give WRONG results."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from lotus_amd import _capi
_capi.load(os.path.join(ROOT, "lotus_amd", "liblotus_hip_tuning.so"))
from lotus_amd.backend import HipBackend
from powermon import PowerMonitor
be = HipBackend("cuda:0")
g = torch.Generator(device=be.device); g.manual_seed(5)
def unit(n, d):
    out = torch.empty((n, d), dtype=torch.float16, device=be.device)
    for r0 in range(0, n, 1 << 18):
        r1 = min(n, r0 + (1 << 18))
        out[r0:r1] = torch.nn.functional.normalize(torch.randn((r1 - r0, d), generator=g, device=be.device), dim=1).half()
    return out
cb = be.pack(unit(1_000_000, 768), _capi.PACK_F16)
NQ = int(os.environ.get("RJ_ABLATE_NQ", "32768"))
cq = be.pack(unit(NQ, 768), _capi.PACK_F16)
os.environ["LVS_RQ_CHUNK"] = str(max(4096, NQ))
os.environ["LVS_RQ_JOIN"] = "1"
os.environ["LVS_RJ"] = "1"
ABL = ((0, "full"), (512, "no publication (atomicMax)"), (128, "no drains (visits stay)"), (2, "no epilogue"))
for mode, xbar in (("0", "1"),):
    os.environ["LVS_RQ_MODE"] = mode
    os.environ["LVS_RQ_XBAR"] = xbar
    for dbg, what in ABL:
        os.environ["LVS_RQ_DEBUG"] = str(dbg)
        be.search_keys(cb, cq, 10, 0); be.synchronize()
        be.timing_enable(True)
        with PowerMonitor(skip=0.1) as pm:
            for _ in range(max(4, 150 * 4096 // NQ)):
                be.search_keys(cb, cq, 10, 0)
            be.synchronize()
        tot, cnt = be.timing_read(); be.timing_enable(False)
        ms = tot / max(cnt, 1)
        p = pm.summary()
        ranges = 16 if NQ <= 4096 else 256 // (NQ // 256)
        units = 1_000_000 / ranges / 32 * 2
        cyc = ms * 1e-3 / units * (p.get("sclk_mhz") or 0) * 1e6
        print(f"mode {mode} xbar {xbar} {what:26s}: kernel {ms:6.3f} ms  sclk {p.get('sclk_mhz')} MHz  {p.get('power_w')} W  ~{cyc:5.0f} cycles per unit", flush=True)

# cycle / event counters of the instrumented instantiation (LVS_RQ_DEBUG=32)
import ctypes
os.environ["LVS_RQ_DEBUG"] = "32"
buf = (ctypes.c_ulonglong * 16)()
be.search_keys(cb, cq, 10, 0); be.synchronize()
be.lib.lvs_rj_debug_read(buf)
be.search_keys(cb, cq, 10, 0); be.synchronize()
be.lib.lvs_rj_debug_read(buf)
v = list(buf)
waves, blocks = max(v[8], 1), max(v[7], 1)
print(f"stamps (per wave-block): epilogue {v[6] / blocks:.0f} cycles of which visits {v[0] / blocks:.0f} ({v[3] / blocks:.3f} visits of {v[0] / max(v[3], 1):.0f} cycles), "
      f"drains {v[1] / blocks:.0f} ({v[4] / blocks:.3f} drains of {v[1] / max(v[4], 1):.0f} cycles, {v[5] / max(v[4], 1):.1f} candidates each = {v[1] / max(v[5], 1):.0f} cycles per candidate); "
      f"vmcnt wait + barrier {v[2] / blocks:.0f} per block; candidates per wave-block {v[5] / blocks:.3f}; waves {waves}, blocks per wave {blocks / waves:.0f}", flush=True)
