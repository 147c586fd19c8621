#!/bin/bash
cd $GRAFT_REPO_ROOT
for S in 100000x1000000 100000x125000 10000x1000000 65536x131072; do echo "== K=10 $S"; QB_REPS=4 timeout 100 python tools/quick_bench.py $S 2>&1 | grep -E "TFLOP"; done
