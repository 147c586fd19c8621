#!/bin/bash
cd $GRAFT_REPO_ROOT
for ns in 6 11 16 21 32; do echo "== NSLAB=$ns"; LVS_NSLAB=$ns QB_REPS=4 python tools/quick_bench.py 100000x1000000 2>&1 | grep -E "TFLOP"; done
