"""Development aid: the CPU comparator (oracle/c/lvs_blas_twin.c) under different OpenMP thread counts / placements on the
GPU box's host (2 x EPYC 9575F).  usage: python tools/cpu_twin_probe.py   (spawns one subprocess per setting)"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CODE = r'''
import sys, time, numpy as np
sys.path.insert(0, %r)
import benchdata
from oracle import blas_twin
xb = benchdata.corpus(3, 250_000, 768).astype(np.float32)
xq = benchdata.queries(3, xb, 4096)[0].astype(np.float32)
th = int(sys.argv[1])
blas_twin.flat_search_c(xb[:65536], xq[:256], 10, threads=th)
t0 = time.perf_counter(); blas_twin.flat_search_c(xb, xq, 10, threads=th); dt = time.perf_counter() - t0
print(f"threads={th} {dt:.2f} s  {2*4096*250000*768/dt/1e12:.2f} TFLOP/s", flush=True)
''' % ROOT
for th, env in ((256, {}), (128, {}), (128, {"OMP_PROC_BIND": "spread", "OMP_PLACES": "cores"}), (256, {"OMP_PROC_BIND": "close", "OMP_PLACES": "threads"}),
                (64, {"OMP_PROC_BIND": "spread", "OMP_PLACES": "cores"})):
    e = dict(os.environ); e.update(env)
    r = subprocess.run([sys.executable, "-c", CODE, str(th)], env=e, capture_output=True, text=True, timeout=300)
    print(env, (r.stdout.strip() or r.stderr[-300:]), flush=True)
