#!/bin/bash
# Board power / clocks sampled with rocm-smi while the headline search runs back to back (development aid):
#   tools/power_probe.sh <out-dir>      -> <out-dir>/power.log (one rocm-smi sample per ~0.2 s), <out-dir>/power_run.log
# Shows whether the tile kernel runs at the board's power cap (the clock the kernel reaches is then set by the cap).
set -u
OUT=${1:-gpurun_out/power}
mkdir -p "$OUT"
rocm-smi --showmaxpower --showpower --showclocks > "$OUT/power_idle.log" 2>&1
python - > "$OUT/power_run.log" 2>&1 <<'PY' &
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from lotus_amd import _capi
from lotus_amd.backend import HipBackend
be = HipBackend("cuda:0")
g = torch.Generator(device=be.device); g.manual_seed(1)
nq, nb, d = 100000, 1000000, 768
xb = torch.nn.functional.normalize(torch.randn((nb, d), generator=g, device=be.device), dim=1).to(torch.float16)
xq = torch.nn.functional.normalize(torch.randn((nq, d), generator=g, device=be.device), dim=1).to(torch.float16)
cb, cq = be.pack(xb, _capi.PACK_F16), be.pack(xq, _capi.PACK_F16)
for phase, (b, q) in (("random", (cb, cq)),):
    be.search_keys(b, q, 10, 0); be.synchronize()
    t0 = time.time(); n = 0
    while time.time() - t0 < 6.0:
        be.search_keys(b, q, 10, 0); n += 1
    be.synchronize()
    print(phase, "calls", n, "ms per call", (time.time() - t0) / n * 1e3, flush=True)
PY
PID=$!
sleep 14   # import + data generation + warm-up
for i in $(seq 1 20); do
  rocm-smi --showpower --showclocks 2>&1 | grep -E "Power|sclk|mclk|fclk" | tr '\n' ' ' >> "$OUT/power.log"; echo >> "$OUT/power.log"
  sleep 0.2
done
wait $PID
