#!/bin/bash
# usage: tools/pmc_mid.sh <tag> <nq> [kernel substring]  - SQ / TCC / fetch counters of one kernel in tools/midbatch_trace.py
# (separate --pmc passes with --kernel-trace only; every pass under a timeout: rocprofv3 can hang after an abort)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
tag=$1; nq=$2; kn=${3:-lvs_rj_kernel}
for c in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE"; do
 n=$(echo $c | cut -d" " -f1)
 timeout -k 5 150 rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pm_${tag}_$n -o b -- python tools/midbatch_trace.py $nq > gpurun_out/pm_${tag}_$n.log 2>&1
 python - <<PY
import csv,collections
agg=collections.defaultdict(list); dur=[]
try:
    for r in csv.DictReader(open("gpurun_out/pm_${tag}_$n/b_counter_collection.csv")):
        if "$kn" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"])); dur.append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e6)
    print("nq $nq", {k:"%.5g"%(sum(v)/len(v)) for k,v in agg.items()}, "ms=%.4f"%(sum(dur)/max(1,len(dur))))
except Exception as e: print("ERR", e)
PY
done
