#!/bin/bash
cd $GRAFT_REPO_ROOT
for w in 256 512; do for S in 1x1000000 32x1000000; do echo "== W8 WGS=$w $S"; LVS_STREAM_WGS=$w QB_REPS=6 timeout 100 python tools/quick_bench.py $S 2>&1 | grep -E "TFLOP"; done; done
