"""T_call with the two stagings of a large host call (development aid): shares (0.1, 0.9) against stages cut at the search's own
chunk boundaries (HipBackend.CALL_CHUNK).  100 k x 1 M, median of 5, alternating."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import benchdata
from lotus_amd import _capi
from lotus_amd.backend import HipBackend
from lotus_amd.vs import HipVS, _Resident

be = HipBackend("cuda:0")
n, nq, d, k = 1_000_000, 100_000, 768, 10
xb = benchdata.corpus(benchdata.CFG_JOIN, n, d)
xq, _ = benchdata.queries(benchdata.CFG_JOIN, xb, nq)
corpus = be.pack(xb, _capi.PACK_F16)
vs = HipVS(backend=be, storage="fp16")
vs._resident["b"] = _Resident(vecs=None, packed=corpus, n=n, d=d, lo=0, hi=n)
vs.index_dir = "b"
queries = be.pack(xq, _capi.PACK_F16)
for _ in range(2):
    be.search_keys(corpus, queries, k, 0)
be.synchronize()
t0 = time.perf_counter()
for _ in range(3):
    be.keys_to_result(be.search_keys(corpus, queries, k, 0), 0)
be.synchronize()
print(f"device-resident step: {(time.perf_counter() - t0) / 3 * 1e3:.2f} ms", flush=True)
ref = None
for rnd in range(2):
    for name, chunk in (("shares (0.1, 0.9)", 10 ** 9), ("chunk-aligned stages", 32768)):
        type(be).CALL_CHUNK = chunk
        vs(xq, k)
        ts = []
        for _ in range(5):
            t0 = time.perf_counter()
            out = vs(xq, k)
            ts.append(time.perf_counter() - t0)
        if ref is None:
            ref = out.indices.copy()
        print(f"{name:24s}: T_call median {sorted(ts)[2] * 1e3:.2f} ms (min {min(ts) * 1e3:.2f}); ids equal to the first run: {bool((out.indices == ref).all())}", flush=True)
