"""Development aid: the CPU comparator's rate against the OpenMP thread count on the GPU box's host, one process.
usage: python tools/cpu_twin_threads.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import benchdata
from oracle import blas_twin

xb = benchdata.corpus(3, 262_144, 768).astype(np.float32)
xq = benchdata.queries(3, xb, 8192)[0].astype(np.float32)
blas_twin.flat_search_c(xb[:65536], xq[:256], 10)
for th in (32, 64, 96, 128, 192, 256):
    if th > (os.cpu_count() or 1):
        continue
    best = 1e9
    for _ in range(2):
        t0 = time.perf_counter(); blas_twin.flat_search_c(xb, xq, 10, threads=th); best = min(best, time.perf_counter() - t0)
    print(f"threads={th:3d} {best:.2f} s  {2 * 8192 * 262144 * 768 / best / 1e12:.2f} TFLOP/s", flush=True)
