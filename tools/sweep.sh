#!/bin/bash
cd $GRAFT_REPO_ROOT
S=65536x131072
for k in 2 5 10 15; do echo "== K=$k $S"; QB_K=$k QB_REPS=4 timeout 100 python tools/quick_bench.py $S 2>&1 | grep -E "TFLOP"; done
for S in 100000x1000000 10000x1000000 100000x125000 1000x10000; do echo "== K=10 $S"; QB_REPS=4 timeout 100 python tools/quick_bench.py $S 2>&1 | grep -E "TFLOP"; done
echo "== K=10 no slow path 100000x1000000"; LVS_DEBUG_HOT=2 QB_K=10 QB_REPS=4 python tools/quick_bench.py 100000x1000000 2>&1 | grep -E "TFLOP"
