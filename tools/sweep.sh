#!/bin/bash
cd $GRAFT_REPO_ROOT
for m in 0 3 2; do echo "== 100kx1M debug=$m"; LVS_DEBUG_HOT=$m QB_REPS=3 python tools/quick_bench.py 100000x1000000 2>&1 | grep -E "TFLOP"; done
for ns in 3 5 7 16; do echo "== NSLAB=$ns 100kx1M"; LVS_NSLAB=$ns QB_REPS=3 python tools/quick_bench.py 100000x1000000 2>&1 | grep -E "TFLOP"; done
