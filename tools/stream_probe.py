"""Where the time of the small-batch streaming kernel goes (TUNING build: LVS_STREAM_DEBUG 1 = no MFMA / B reads,
2 = no block epilogue; LVS_STREAM_SEED 0 = no sample pass).  Development aid.  usage: python tools/stream_probe.py [rows]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from lotus_amd import _capi
_capi.load(os.path.join(ROOT, "lotus_amd", "liblotus_hip_tuning.so"))
from lotus_amd.backend import HipBackend

be = HipBackend("cuda:0")
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
d, k = 768, 10
g = torch.Generator(device=be.device); g.manual_seed(1)
xb = torch.nn.functional.normalize(torch.randn((nb, d), generator=g, device=be.device), dim=1).to(torch.float16)
j = torch.randint(0, nb, (96,), generator=g, device=be.device)
xq = torch.nn.functional.normalize(0.7 * xb[j].float() + 0.7 * torch.nn.functional.normalize(torch.randn((96, d), generator=g, device=be.device), dim=1), dim=1).to(torch.float16)
cb, cq = be.pack(xb, _capi.PACK_F16), be.pack(xq, _capi.PACK_F16)

def wall_us(q, reps=20):
    for _ in range(3):
        be.search_keys(cb, q, k, 0)
    be.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        be.search_keys(cb, q, k, 0)
    be.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6

for nq in (1, 8, 32, 64, 96):
    q = be.slice_rows(cq, 0, nq)
    out = []
    for seed in ("1", "0"):
        for dbg in ("0", "1", "2"):
            os.environ["LVS_STREAM_SEED"] = seed
            os.environ["LVS_STREAM_DEBUG"] = dbg
            out.append(f"seed{seed}/dbg{dbg} {wall_us(q):7.1f}")
    print(f"nq={nq:3d} x {nb}: " + "  ".join(out) + "  (us per call, wall)", flush=True)
