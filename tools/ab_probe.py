"""Same-box A/B of two builds of the library (development aid): python tools/ab_probe.py <path of liblotus_hip*.so>
(the other build: compile another checkout's lotus_amd/csrc - same ABI version - and copy its library into lotus_amd/ under
another name, e.g. liblotus_hip_prevtile.so, which `tools/r04_run.sh <tag> ab` alternates with the shipped one; boxes of the
pool differ by more than most kernel changes, so only runs of one gpurun call compare)
fp16 join 100 k x 1 M, fp16 / fp32 10 k x 1 M, fp32 join 100 k x 1 M (d = 768, k = 10, IP), ms per call by device events."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lotus_amd import _capi

lib = _capi.load(os.path.abspath(sys.argv[1]))
_capi._lib = lib
from lotus_amd.backend import HipBackend

be = HipBackend("cuda:0")
n, nq, d, k = 1_000_000, 100_000, 768, 10
g = torch.Generator(device=be.device); g.manual_seed(3)
xb = torch.nn.functional.normalize(torch.randn((n, d), generator=g, device=be.device), dim=1)
j = torch.randint(0, n, (nq,), generator=g, device=be.device)
xq = torch.nn.functional.normalize(0.7 * xb[j] + 0.7 * torch.nn.functional.normalize(torch.randn((nq, d), generator=g, device=be.device), dim=1), dim=1)
c32, q32 = be.pack(xb, _capi.PACK_SPLIT), be.pack(xq, _capi.PACK_SPLIT)
c16, q16 = be.pack(xb.half(), _capi.PACK_F16), be.pack(xq.half(), _capi.PACK_F16)
del xb, xq
q16s, q32s = be.slice_rows(q16, 0, 10_000), be.slice_rows(q32, 0, 10_000)


def timed(fn, reps):
    fn(); be.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
    ev[0].record()
    for i in range(reps):
        fn(); ev[i + 1].record()
    be.synchronize()
    ms = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(reps))
    return ms[len(ms) // 2], ms[0]


out = [os.path.basename(sys.argv[1])]
for name, c, q, reps in (("fp16 100k", c16, q16, 5), ("fp16 10k", c16, q16s, 9), ("fp32 10k", c32, q32s, 9), ("fp32 100k", c32, q32, 3),
                         ("fp16 100k", c16, q16, 5)):
    st = {}
    med, best = timed(lambda: be.search_keys(c, q, k, 0, stats=st), reps)
    out.append(f"{name}: {med:.2f} (min {best:.2f})" + (f" open {st.get('uncertified', 0)}/{st.get('queries', 0)}" if st else ""))
print(" | ".join(out), flush=True)
