#!/bin/bash
cd $GRAFT_REPO_ROOT
echo "== normal"; QB_REPS=4 python tools/quick_bench.py 100000x1000000 2>&1 | grep -E "TFLOP"
echo "== noslow"; LVS_DEBUG_HOT=2 QB_REPS=4 python tools/quick_bench.py 100000x1000000 2>&1 | grep -E "TFLOP"
echo "== k=2"; QB_K=2 QB_REPS=4 python tools/quick_bench.py 100000x1000000 2>&1 | grep -E "TFLOP"
echo "== k=5"; QB_K=5 QB_REPS=4 python tools/quick_bench.py 100000x1000000 2>&1 | grep -E "TFLOP"
./tools/probe_gemm.bin 2>&1 | tail -4 | head -2
