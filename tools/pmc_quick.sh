#!/bin/bash
# usage: tools/pmc_quick.sh <tag> [env assignments...]  - SQ/TCC counters of the tile kernel at 100k x 1M
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
tag=$1; shift
for c in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT" "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE"; do
 n=$(echo $c | cut -d" " -f1)
 env "$@" rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pq_${tag}_$n -o b -- python tools/quick_bench.py 100000x1000000 > gpurun_out/pq_${tag}_$n.log 2>&1
 python - <<PY
import csv,collections
agg=collections.defaultdict(list); dur=[]
try:
    for r in csv.DictReader(open("gpurun_out/pq_${tag}_$n/b_counter_collection.csv")):
        if "lvs_tile" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"])); dur.append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e6)
    print({k:"%.4g"%(sum(v)/len(v)) for k,v in agg.items()}, "ms=%.1f"%(sum(dur)/max(1,len(dur))))
except Exception as e: print("ERR", e)
PY
done
