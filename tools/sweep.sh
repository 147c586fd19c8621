#!/bin/bash
cd $GRAFT_REPO_ROOT
for cfg in "32 1" "8 1" "4 1" "16 1" "8 0"; do set -- $cfg
  echo "== v2 GQ=$1 NT=$2"; LVS_NT=$2 LVS_GQ=$1 python tools/quick_bench.py 100000x1000000 2>&1 | grep -E "TFLOP"
done
