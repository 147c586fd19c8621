"""Round 6, review item 1: variants of the join-scale search (k = 10, d = 768 fp16, 1 M rows) on ONE box, alternating, each with
kernel time (the library's events), wall time per call, mean board power and engine clock over >= 3 s of back-to-back calls, and
the keys compared bit for bit with the list kernel's.  Tuning build (knobs read per call).
   python tools/join_variants_probe.py [--nq 4096,10000,100000] [--secs 3] [--rounds 2] [--variants list,rq32k,default]"""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from lotus_amd import _capi
_capi.load(os.path.join(ROOT, "lotus_amd", "liblotus_hip_tuning.so"))
from lotus_amd.backend import HipBackend
from powermon import PowerMonitor

ap = argparse.ArgumentParser()
ap.add_argument("--nq", default="4096,10000,100000")
ap.add_argument("--nb", type=int, default=1_000_000)
ap.add_argument("--secs", type=float, default=3.0)
ap.add_argument("--rounds", type=int, default=2)
ap.add_argument("--variants", default="list,rq32k,default")
ap.add_argument("--planted", type=int, default=1)
args = ap.parse_args()

VARIANTS = {  # name -> environment of the tuning build
    "default": {},                                              # what the shipped library does: lvs_rj_kernel, chunks of 32 768 queries
    "list": {"LVS_RQ_JOIN": "0", "LVS_RQ_MAXG": "0"},          # the list kernel for everything beyond 256 queries (round 5's path)
    "rq": {"LVS_RQ_JOIN": "1", "LVS_RJ": "0", "LVS_RQ_CHUNK": "4096"},   # lvs_rq_kernel (two waves per SIMD) in chunks of 4 096 queries
    "rq32k": {"LVS_RQ_JOIN": "1", "LVS_RJ": "0", "LVS_RQ_CHUNK": "32768"},
    "rjk": {"LVS_RQ_JOIN": "1", "LVS_RJ": "1", "LVS_RQ_CHUNK": "4096"},  # lvs_rj_kernel (one wave per SIMD) in chunks of 4 096 queries
    "rjk8k": {"LVS_RQ_JOIN": "1", "LVS_RJ": "1", "LVS_RQ_CHUNK": "8192"},
    "rjk16k": {"LVS_RQ_JOIN": "1", "LVS_RJ": "1", "LVS_RQ_CHUNK": "16384"},
    "rjk32k": {"LVS_RQ_JOIN": "1", "LVS_RJ": "1", "LVS_RQ_CHUNK": "32768"},
    "rjk64k": {"LVS_RQ_JOIN": "1", "LVS_RJ": "1", "LVS_RQ_CHUNK": "65536"},
    "min16k": {"LVS_RQ_JOIN_MINROWS": "16384"},                # shortest corpus the chunked path takes
    "min8k": {"LVS_RQ_JOIN_MINROWS": "8192"},
    "d_s4k": {"LVS_RQ_SAMPLE": "4096"},                         # sample rows of the seed pass
    "d_s16k": {"LVS_RQ_SAMPLE": "16384"},
    "d_s32k": {"LVS_RQ_SAMPLE": "32768"},
    "d_e4": {"LVS_RJ_EVERY": "4"},                              # blocks between the waves' common drains
    "d_e8": {"LVS_RJ_EVERY": "8"},
    "d_e32": {"LVS_RJ_EVERY": "32"},
}
KNOBS = sorted({k for v in VARIANTS.values() for k in v})

be = HipBackend("cuda:0")
g = torch.Generator(device=be.device); g.manual_seed(20260930)
d, nb = 768, args.nb

def unit(n):
    out = torch.empty((n, d), dtype=torch.float16, device=be.device)
    for r0 in range(0, n, 1 << 18):
        r1 = min(n, r0 + (1 << 18))
        out[r0:r1] = torch.nn.functional.normalize(torch.randn((r1 - r0, d), generator=g, device=be.device), dim=1).half()
    return out

xb = unit(nb)
cb = be.pack(xb, _capi.PACK_F16)
nq_max = max(int(x) for x in args.nq.split(","))
if args.planted:  # q = normalize(0.7 x[j] + 0.7 u)
    j = torch.randint(0, nb, (nq_max,), generator=g, device=be.device)
    xq = torch.empty((nq_max, d), dtype=torch.float16, device=be.device)
    for r0 in range(0, nq_max, 1 << 16):
        r1 = min(nq_max, r0 + (1 << 16))
        u = torch.nn.functional.normalize(torch.randn((r1 - r0, d), generator=g, device=be.device), dim=1)
        xq[r0:r1] = torch.nn.functional.normalize(0.7 * xb[j[r0:r1]].float() + 0.7 * u, dim=1).half()
else:
    xq = unit(nq_max)
del xb

def setenv(name):
    for k in KNOBS:
        os.environ.pop(k, None)
    os.environ.update(VARIANTS[name])

def one(name, cq, secs):
    setenv(name)
    keys = be.search_keys(cb, cq, 10, 0)
    be.synchronize()
    be.timing_enable(True)
    n = 0
    with PowerMonitor() as pm:
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < secs or n < 3:
            keys = be.search_keys(cb, cq, 10, 0)
            n += 1
            if n % 4 == 0:
                be.synchronize()
        be.synchronize()
        wall = (time.perf_counter() - t0) / n * 1e3
    tot, cnt = be.timing_read()
    be.timing_enable(False)
    return keys, tot / n, cnt / n, wall, pm.summary()

names = [v for v in args.variants.split(",") if v in VARIANTS]
rows = []
for nq in (int(x) for x in args.nq.split(",")):
    cq = be.pack(xq[:nq].contiguous(), _capi.PACK_F16)
    flop = 2.0 * nq * nb * d
    ref = None
    for rnd in range(args.rounds):
        for name in names:
            keys, kms, launches, wall, pw = one(name, cq, args.secs)
            if name == "list" and ref is None:
                ref = keys.clone()
            same = None if ref is None else bool(torch.equal(keys, ref))
            tf = flop / (kms * 1e-3) / 1e12
            row = {"nq": nq, "variant": name, "round": rnd, "kernel_ms": round(kms, 4), "launches": launches, "wall_ms": round(wall, 4),
                   "tflops": round(tf, 1), "frac": round(tf / 2500, 4), "keys_equal_list": same, **{k: pw.get(k) for k in ("power_w", "sclk_mhz", "samples", "source")}}
            if pw.get("power_w"):
                row["gflop_per_j"] = round(flop / (wall * 1e-3) / pw["power_w"] / 1e9, 2)
            rows.append(row)
            print(json.dumps(row), flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(rows, open(os.path.join(ROOT, "gpurun_out", "join_variants_probe.json"), "w"), indent=1)
