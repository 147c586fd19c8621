cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for gq in 32 8; do
for c in "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES"; do
 n=$(echo $c | cut -d" " -f1)
 LVS_GQ=$gq QB_REPS=2 rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pg_${gq}_$n -o b -- python tools/quick_bench.py 100000x1000000 > gpurun_out/pg_${gq}_$n.log 2>&1
 python - <<PY
import csv,collections
agg=collections.defaultdict(list); dur=[]
try:
    for r in csv.DictReader(open("gpurun_out/pg_${gq}_$n/b_counter_collection.csv")):
        if "lvs_tile" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"])); dur.append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e6)
    print("gq=$gq", {k:"%.4g"%(sum(v)/len(v)) for k,v in agg.items()}, "ms=%.1f"%(sum(dur)/max(1,len(dur))))
except Exception as e: print("ERR", e)
PY
done; done
find gpurun_out -name "*.db" -delete
