"""Quick single-GPU timing of the search kernel at BASELINE shapes (development aid, not the judged bench)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from lotus_amd.backend import HipBackend
from lotus_amd import _capi

be = HipBackend("cuda:0")
print(torch.cuda.get_device_name(0), torch.cuda.get_device_properties(0).multi_processor_count, "CUs", flush=True)
d, k = 768, int(os.environ.get("QB_K", "10"))
shapes = [(10000, 1_000_000), (100000, 1_000_000)] if len(sys.argv) < 2 else [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]]
g = torch.Generator(device=be.device); g.manual_seed(1)
nmax = max(s[1] for s in shapes); qmax = max(s[0] for s in shapes)
xb = torch.nn.functional.normalize(torch.randn((nmax, d), generator=g, device=be.device), dim=1).to(torch.float16)
j = torch.randint(0, nmax, (qmax,), generator=g, device=be.device)
xq = torch.nn.functional.normalize(0.7 * xb[j].float() + 0.7 * torch.nn.functional.normalize(torch.randn((qmax, d), generator=g, device=be.device), dim=1), dim=1).to(torch.float16)
for nq, nb in shapes:
    cb, cq = be.pack(xb[:nb], _capi.PACK_F16), be.pack(xq[:nq], _capi.PACK_F16)
    for it in range(2):
        keys = be.search_keys(cb, cq, k, 0)
    be.synchronize()
    reps = int(os.environ.get("QB_REPS", "6"))
    ks = []
    t0 = time.time()
    for it in range(reps):
        be.timing_enable(True)
        keys = be.search_keys(cb, cq, k, 0)
        be.synchronize()
        tot, cnt = be.timing_read()
        ks.append(tot / max(cnt, 1))
    dt = (time.time() - t0) / reps
    be.timing_enable(False)
    ks.sort()
    fl = 2.0 * nq * nb * d
    kmin, kmed = ks[0], ks[len(ks) // 2]
    print(f"{nq}x{nb}: kernel min {kmin:.2f} ms med {kmed:.2f} ms  {fl/(kmin*1e-3)/1e12:.1f} TFLOP/s (min) {fl/(kmed*1e-3)/1e12:.1f} TFLOP/s (med)  wall {dt*1e3:.2f} ms {nq/dt:.0f} q/s", flush=True)
    D, I = be.keys_to_result(keys, 0)
    print("  planted@1:", float((I[:, 0] == j[:nq]).float().mean()), flush=True)
