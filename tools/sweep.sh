#!/bin/bash
cd $GRAFT_REPO_ROOT
for L in 0 1; do for S in 100000x1000000 100000x125000; do echo "== LAYOUT=$L $S"; LVS_LAYOUT=$L QB_REPS=4 timeout 100 python tools/quick_bench.py $S 2>&1 | grep -E "TFLOP|planted"; done; done
echo "== LAYOUT=1 noslow"; LVS_LAYOUT=1 LVS_DEBUG_HOT=2 QB_REPS=3 timeout 100 python tools/quick_bench.py 100000x1000000 2>&1 | grep -E "TFLOP"
echo "== LAYOUT=1 K=1"; LVS_LAYOUT=1 QB_K=1 QB_REPS=3 timeout 100 python tools/quick_bench.py 100000x1000000 2>&1 | grep -E "TFLOP"
