"""fp32 embeddings (hi|lo rows), 10 k x 1 M x 768, k = 10 (development aid): (1) which launches the certified one-pass search
consists of (device time per C-ABI call, from events recorded around every call), (2) does the hi pass care that the hi parts
sit at the hi|lo rows' 3 072-byte stride?  - the same call against a CONTIGUOUS copy of the hi parts (1 536-byte rows), with
the same error bound, so the lists see the same band."""
import os, sys, time, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lotus_amd import _capi
from lotus_amd.backend import HipBackend, PackedRows

be = HipBackend("cuda:0")
n, nq, d, k = 1_000_000, 10_000, 768, 10
g = torch.Generator(device=be.device); g.manual_seed(3)
xb = torch.nn.functional.normalize(torch.randn((n, d), generator=g, device=be.device), dim=1)
j = torch.randint(0, n, (nq,), generator=g, device=be.device)
xq = torch.nn.functional.normalize(0.7 * xb[j] + 0.7 * torch.nn.functional.normalize(torch.randn((nq, d), generator=g, device=be.device), dim=1), dim=1)
c32, q32 = be.pack(xb, _capi.PACK_SPLIT, exp="auto"), None
q32 = be.pack(xq, _capi.PACK_SPLIT, exp=c32.exp)
c16, q16 = be.pack(xb.half(), _capi.PACK_F16), be.pack(xq.half(), _capi.PACK_F16)
del xb, xq
dpad = int(c32.rows.shape[1]) // 2
hi = PackedRows(rows=c32.rows[:, :dpad].contiguous(), norms=c32.norms, n=c32.n, d=c32.d, mode=_capi.PACK_F16, exp=c32.exp)
E = be.lo_norm_max(c32)
orig_lo = be.lo_norm_max
be.lo_norm_max = lambda pk: E if pk is hi else orig_lo(pk)

rec = []
orig_c = be._c
def timed_c(name, *args):
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record(); orig_c(name, *args); e.record()
    rec.append((name, s, e))

def run(tag, fn, reps=5):
    for _ in range(2):
        fn()
    be.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    be.synchronize()
    wall = (time.perf_counter() - t0) / reps * 1e3
    be._c = timed_c
    rec.clear()
    st = {}
    for _ in range(reps):
        fn(st)
    be.synchronize()
    be._c = orig_c
    by = collections.OrderedDict()
    for name, s, e in rec:
        by[name] = by.get(name, 0.0) + s.elapsed_time(e) / reps
    print(f"{tag}: {wall:.2f} ms per call (wall, untimed run)  {st}", flush=True)
    for name, ms in by.items():
        print(f"    {name:36s} {ms:7.3f} ms", flush=True)

for rnd in range(2):
    run("fp16 10k x 1M", lambda st=None: be.search_keys(c16, q16, k, 0))
    run("fp32 one-pass, hi|lo rows (stride 3072 B)", lambda st=None: be._search_keys_certified(c32, q32, k, 0, 0, st))
    run("fp32 one-pass, hi parts contiguous (1536 B)", lambda st=None: be._search_keys_certified(hi, q32, k, 0, 0, st))
