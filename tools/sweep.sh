#!/bin/bash
cd $GRAFT_REPO_ROOT
echo "== K=1 normal"; QB_K=1 QB_REPS=3 python tools/quick_bench.py 100000x1000000 2>&1 | grep -E "TFLOP"
echo "== K=1 no-vmcnt-wait"; LVS_DEBUG_HOT=4 QB_K=1 QB_REPS=3 python tools/quick_bench.py 100000x1000000 2>&1 | grep -E "TFLOP"
