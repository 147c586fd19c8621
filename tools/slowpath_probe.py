"""Slow-path accounting of the tile kernel with the TUNING build (liblotus_hip_tuning.so: LVS_* knobs + event counters).
Development aid - never part of the product path.  usage: python tools/slowpath_probe.py [QxN ...]
Per shape: kernel ms as shipped, with the slow path skipped (LVS_DEBUG_HOT=2: wrong results, timing only), with hits
scanned but not inserted (=3), and the event counters of one counted call (block visits, insertions, cycles)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from lotus_amd import _capi
_capi.load(os.path.join(ROOT, "lotus_amd", "liblotus_hip_tuning.so"))  # explicit: the tuning build
from lotus_amd.backend import HipBackend

be = HipBackend("cuda:0")
d, k = 768, int(os.environ.get("QB_K", "10"))
shapes = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]] or [(100000, 125000), (100000, 1000000)]
g = torch.Generator(device=be.device); g.manual_seed(1)
nmax = max(s[1] for s in shapes); qmax = max(s[0] for s in shapes)
xb = torch.nn.functional.normalize(torch.randn((nmax, d), generator=g, device=be.device), dim=1).to(torch.float16)
j = torch.randint(0, nmax, (qmax,), generator=g, device=be.device)
xq = torch.nn.functional.normalize(0.7 * xb[j].float() + 0.7 * torch.nn.functional.normalize(
    torch.randn((qmax, d), generator=g, device=be.device), dim=1), dim=1).to(torch.float16)

def kernel_ms(cb, cq, reps=4):
    for _ in range(2):
        be.search_keys(cb, cq, k, 0)
    be.synchronize()
    ts = []
    for _ in range(reps):
        be.timing_enable(True)
        be.search_keys(cb, cq, k, 0)
        be.synchronize()
        tot, cnt = be.timing_read()
        ts.append(tot / max(cnt, 1))
    be.timing_enable(False)
    return min(ts)

for nq, nb in shapes:
    cb, cq = be.pack(xb[:nb], _capi.PACK_F16), be.pack(xq[:nq], _capi.PACK_F16)
    fl = 2.0 * nq * nb * d
    res = {}
    for tag, env in (("shipped", {}), ("no_slow_path", {"LVS_DEBUG_HOT": "2"}), ("no_insertions", {"LVS_DEBUG_HOT": "3"})):
        for kk, v in env.items():
            os.environ[kk] = v
        res[tag] = kernel_ms(cb, cq)
        for kk in env:
            del os.environ[kk]
    print(f"{nq}x{nb}: " + "  ".join(f"{t} {ms:.2f} ms ({fl / (ms * 1e-3) / 1e12:.0f} TF)" for t, ms in res.items()), flush=True)
    os.environ["LVS_COUNT"] = "1"
    be.search_keys(cb, cq, k, 0)
    be.synchronize()
    del os.environ["LVS_COUNT"]
