"""Small and medium batches (1 .. 10 000 queries x 1 M rows x 768 fp16, k = 10): main-kernel us (library HIP events) and wall us
per call of the shipped dispatch against its alternatives (TUNING build knobs), results compared bit for bit.
Development aid.  usage: python tools/small_batch_sweep.py [rows] [trace]   ("trace": few calls only, for rocprofv3)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from lotus_amd import _capi
trace = "trace" in sys.argv[1:]
if not trace:
    _capi.load(os.path.join(ROOT, "lotus_amd", "liblotus_hip_tuning.so"))
from lotus_amd.backend import HipBackend

be = HipBackend("cuda:0")
args = [a for a in sys.argv[1:] if a != "trace"]
nb = int(args[0]) if args else 1_000_000
d, k = 768, 10
g = torch.Generator(device=be.device); g.manual_seed(1)
xb = torch.nn.functional.normalize(torch.randn((nb, d), generator=g, device=be.device), dim=1).to(torch.float16)
j = torch.randint(0, nb, (256,), generator=g, device=be.device)
xq = torch.nn.functional.normalize(0.7 * xb[j].float() + 0.7 * torch.nn.functional.normalize(torch.randn((256, d), generator=g, device=be.device), dim=1), dim=1).to(torch.float16)
cb, cq = be.pack(xb, _capi.PACK_F16), be.pack(xq, _capi.PACK_F16)

def run(q, reps=20):
    for _ in range(3):
        be.keys_to_result(be.search_keys(cb, q, k, 0), 0)
    be.synchronize()
    be.timing_enable(True)
    t0 = time.perf_counter()
    for _ in range(reps):
        be.keys_to_result(be.search_keys(cb, q, k, 0), 0)
    be.synchronize()
    wall = (time.perf_counter() - t0) / reps * 1e6
    tot, cnt = be.timing_read(); be.timing_enable(False)
    return tot / max(cnt, 1) * 1e3, wall

if trace:
    for nq in (1, 32, 64, 96, 128, 192, 256):
        run(be.slice_rows(cq, 0, nq), reps=5)
    sys.exit(0)
KNOBS = ("LVS_STREAM_SEED", "LVS_STREAM_MAXQ_RT", "LVS_STREAM_MAXG", "LVS_TILE_SEED")
VARIANTS = (("shipped", {}),                                    # stream kernel up to one sibling group, seeded list kernel beyond
            ("stream-siblings", {"LVS_STREAM_MAXG": "4"}),      # round 3's first form: 2-4 sibling groups per corpus range
            ("list-seeded", {"LVS_STREAM_MAXQ_RT": "1"}),       # the list (tile) kernel with seeded thresholds at every size
            ("list-cold", {"LVS_STREAM_MAXQ_RT": "1", "LVS_TILE_SEED": "0"}))
sizes = (1, 32, 64, 96, 128, 160, 192, 256, 512, 1024, 2048, 4096, 10000)
nmax = max(sizes)
j = torch.randint(0, nb, (nmax,), generator=g, device=be.device)
xq = torch.nn.functional.normalize(0.7 * xb[j].float() + 0.7 * torch.nn.functional.normalize(torch.randn((nmax, d), generator=g, device=be.device), dim=1), dim=1).to(torch.float16)
cq = be.pack(xq, _capi.PACK_F16)
for nq in sizes:
    q = be.slice_rows(cq, 0, nq)
    out, ref = [], None
    for tag, env in VARIANTS:
        for kk in KNOBS:
            os.environ.pop(kk, None)
        os.environ.update(env)
        if nq > 256 and tag == "stream-siblings":
            continue
        kus, wus = run(q, reps=20 if nq <= 1024 else 5)
        keys = be.search_keys(cb, q, k, 0)
        same = "" if ref is None else (" =" if torch.equal(keys, ref) else " DIFFERENT")
        ref = keys if ref is None else ref
        out.append(f"{tag}: kernel {kus:7.1f} wall {wus:7.1f}{same}")
    print(f"nq={nq:5d} x {nb}: " + "   ".join(out) + "   (us)", flush=True)
