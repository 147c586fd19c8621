#!/bin/bash
# rocprofv3 kernel trace + separate --pmc passes of tools/rq_pmc_workload.py -> gpurun_out/<tag>/rq_pmc.txt
tag=${1:-rqpmc}
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
out=gpurun_out/$tag; mkdir -p $out
W="python tools/rq_pmc_workload.py"
timeout -k 5 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o k -- $W > $out/trace.log 2>&1
for c in "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" \
         "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL" "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  n=$(echo $c | cut -d" " -f1)
  timeout -k 5 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/pmc_$n -o k -- $W > $out/pmc_$n.log 2>&1
done
find $out -name "*.db" -delete
python - "$out" <<'PY' | tee $out/rq_pmc.txt
import csv, glob, sys, collections
out = sys.argv[1]
for p in glob.glob(out + "/trace/*kernel_stats.csv"):
    for r in csv.DictReader(open(p)):
        if "lvs_rq" in r["Name"] or "lvs_tile" in r["Name"] or "stream" in r["Name"]:
            print(r["Name"][:90], r["Calls"], "avg ns", r["AverageNs"])
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for p in glob.glob(out + "/pmc_*/k_counter_collection.csv"):
    for r in csv.DictReader(open(p)):
        if "lvs_rq_kernel" in r["Kernel_Name"]:
            agg[r["Kernel_Name"][:80]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in agg.items():
    print(k)
    for c, v in sorted(cs.items()):
        print(f"   {c:28s} {sum(v)/len(v):16.1f}  (n={len(v)})")
PY
