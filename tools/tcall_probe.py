"""T_call anatomy (development aid): HipVS.__call__(host fp16 [Q, d]) -> host (D, I) at 100 k x 1 M against the device-resident
search, for several stagings of the queries (HipBackend.CALL_PIPELINE)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import benchdata
from lotus_amd import _capi
from lotus_amd.backend import HipBackend
from lotus_amd.vs import HipVS, _Resident

be = HipBackend("cuda:0")
n, nq, d, k = 1_000_000, 100_000, 768, 10
xb = benchdata.corpus(benchdata.CFG_JOIN, n, d)
xq, _ = benchdata.queries(benchdata.CFG_JOIN, xb, nq)
corpus = be.pack(xb, _capi.PACK_F16)
queries = be.pack(xq, _capi.PACK_F16)


def dev_ms(q, reps=3):
    for _ in range(2):
        be.search_keys(corpus, q, k, 0)
    be.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        be.keys_to_result(be.search_keys(corpus, q, k, 0), 0)
    be.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


print(f"device-resident 100k: {dev_ms(queries):.2f} ms", flush=True)
for cut in (10_000, 20_000, 30_000, 50_000):
    a, b = dev_ms(be.slice_rows(queries, 0, cut)), dev_ms(be.slice_rows(queries, cut, nq))
    print(f"device-resident split {cut} + {nq - cut}: {a:.2f} + {b:.2f} = {a + b:.2f} ms", flush=True)
# raw transfers
t0 = time.perf_counter(); dq = be._h2d(xq); be.synchronize(); print(f"h2d 154 MB through the ring: {(time.perf_counter()-t0)*1e3:.2f} ms", flush=True)
t0 = time.perf_counter(); dq = be._h2d(xq); be.synchronize(); print(f"h2d again: {(time.perf_counter()-t0)*1e3:.2f} ms", flush=True)
t0 = time.perf_counter(); dq2 = torch.from_numpy(xq).to(be.device); be.synchronize(); print(f"h2d plain .to(): {(time.perf_counter()-t0)*1e3:.2f} ms", flush=True)
vs = HipVS(backend=be, storage="fp16")
vs._resident["b"] = _Resident(vecs=None, packed=corpus, n=n, d=d, lo=0, hi=n)
vs.index_dir = "b"


def tcall(reps=5):
    vs(xq, k)
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); vs(xq, k); ts.append(time.perf_counter() - t0)
    return sorted(ts)[len(ts) // 2] * 1e3


for pipe in ((1.0,), (0.2, 0.8), (0.1, 0.9), (0.3, 0.7), (0.5, 0.5), (0.1, 0.8, 0.1), (0.2, 0.6, 0.2), (0.25, 0.25, 0.25, 0.25)):
    be.CALL_PIPELINE = pipe
    print(f"T_call staged {pipe}: {tcall():.2f} ms", flush=True)
be.CALL_PIPELINE_MIN_QUERIES = 1 << 40
print(f"T_call unstaged: {tcall():.2f} ms", flush=True)
