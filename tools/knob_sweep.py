"""Sweep LVS_* planner knobs of the TUNING build at a given shape (development aid).
usage: python tools/knob_sweep.py QxN KNOB v1 v2 ...        e.g.  100000x1000000 LVS_NSLAB 21 25 29 33
       python tools/knob_sweep.py QxN SET A=1,B=2 A=3 ...   (several knobs per setting)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from lotus_amd import _capi
_capi.load(os.path.join(ROOT, "lotus_amd", "liblotus_hip_tuning.so"))
from lotus_amd.backend import HipBackend

be = HipBackend("cuda:0")
nq, nb = (int(v) for v in sys.argv[1].split("x"))
knob, vals = sys.argv[2], sys.argv[3:]
d, k = 768, 10
g = torch.Generator(device=be.device); g.manual_seed(1)
xb = torch.nn.functional.normalize(torch.randn((nb, d), generator=g, device=be.device), dim=1).to(torch.float16)
j = torch.randint(0, nb, (nq,), generator=g, device=be.device)
xq = torch.nn.functional.normalize(0.7 * xb[j].float() + 0.7 * torch.nn.functional.normalize(torch.randn((nq, d), generator=g, device=be.device), dim=1), dim=1).to(torch.float16)
cb, cq = be.pack(xb, _capi.PACK_F16), be.pack(xq, _capi.PACK_F16)

def kernel_ms(reps=4):
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
    return min(ts), sorted(ts)[len(ts) // 2]

fl = 2.0 * nq * nb * d
touched = set()
for rnd in range(2):  # two interleaved rounds: box drift shows up as a difference between them
    for v in ["default"] + vals:
        for name in touched:
            os.environ.pop(name, None)
        if v != "default":
            for kv in (v.split(",") if knob == "SET" else [f"{knob}={v}"]):
                name, val = kv.split("=")
                os.environ[name] = val
                touched.add(name)
        mn, med = kernel_ms()
        print(f"{sys.argv[1]} {knob}={v:34s} min {mn:8.2f} ms  med {med:8.2f} ms  {fl / (mn * 1e-3) / 1e12:7.1f} TFLOP/s", flush=True)
