#!/bin/bash
cd $GRAFT_REPO_ROOT
for S in 33x1000000 100x1000000 300x1000000 600x1000000 1000x1000000 1500x1000000 3000x1000000; do echo "== $S"; QB_REPS=4 timeout 100 python tools/quick_bench.py $S 2>&1 | grep -E "TFLOP"; done
