"""Seed-sample size of the grouped lvs_rq_kernel launch (LVS_RQ_SAMPLE, tuning build): per call, 1 M x 768 fp16, k = 10."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from lotus_amd import _capi
_capi.load(os.path.join(ROOT, "lotus_amd", "liblotus_hip_tuning.so"))
from lotus_amd.backend import HipBackend

be = HipBackend("cuda:0")
g = torch.Generator(device=be.device); g.manual_seed(11)

def unit(n, d):
    out = torch.empty((n, d), dtype=torch.float16, device=be.device)
    for r0 in range(0, n, 1 << 18):
        r1 = min(n, r0 + (1 << 18))
        out[r0:r1] = torch.nn.functional.normalize(torch.randn((r1 - r0, d), generator=g, device=be.device), dim=1).half()
    return out

xb = unit(1_000_000, 768); cb = be.pack(xb, _capi.PACK_F16); del xb
xq = unit(4096, 768)
os.environ["LVS_RQ"] = "1"
for rnd in range(2):
    for nq in (512, 1024, 2048, 4096):
        cq = be.pack(xq[:nq].contiguous(), _capi.PACK_F16)
        line = f"{nq:5d} q:"
        for sample in (65536, 32768, 16384, 8192, 4096):
            os.environ["LVS_RQ_SAMPLE"] = str(sample)
            be.search_keys(cb, cq, 10, 0); be.synchronize()
            be.timing_enable(True)
            t0 = time.perf_counter()
            for _ in range(10):
                be.search_keys(cb, cq, 10, 0)
            be.synchronize()
            wall = (time.perf_counter() - t0) / 10 * 1e3
            tot, cnt = be.timing_read(); be.timing_enable(False)
            line += f"  s{sample // 1024}k kernel {tot / max(cnt, 1):6.3f} call {wall:6.3f}"
        print(line, flush=True)
