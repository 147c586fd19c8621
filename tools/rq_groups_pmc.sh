#!/bin/bash
# rocprofv3 kernel trace + separate --pmc passes of tools/rq_pmc_workload.py at 256 / 512 / 1024 / 4096 queries (query groups of
# lvs_rq_kernel: does a corpus range reach its sibling workgroups through the XCD's L2?) -> gpurun_out/<tag>/rq_groups_pmc.txt
tag=${1:-rqgpmc}
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
export RQ_PMC_NQ=256,512,1024,4096
out=gpurun_out/$tag; mkdir -p $out
W="python tools/rq_pmc_workload.py"
timeout -k 5 200 rocprofv3 --kernel-trace --output-format csv -d $out/trace -o k -- $W > $out/trace.log 2>&1
for c in "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY"; do
  n=$(echo $c | cut -d" " -f1)
  timeout -k 5 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/pmc_$n -o k -- $W > $out/pmc_$n.log 2>&1
done
find $out -name "*.db" -delete
python - "$out" <<'PY' | tee $out/rq_groups_pmc.txt
import csv, glob, sys, collections
out = sys.argv[1]
sizes = [256, 512, 1024, 4096]
def main_rows(path, name_col):
    rows = [r for r in csv.DictReader(open(path)) if "lvs_rq_kernel" in r[name_col] and "true>" not in r[name_col]]
    return rows
for p in glob.glob(out + "/trace/*kernel_trace.csv"):
    rows = main_rows(p, "Kernel_Name")
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    for i, nq in enumerate(sizes):
        d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in rows[4 * i:4 * i + 4]]
        if d:
            print(f"{nq:5d} queries: lvs_rq_kernel {sum(d) / len(d):.3f} ms (n={len(d)})")
for p in sorted(glob.glob(out + "/pmc_*/k_counter_collection.csv")):
    rows = main_rows(p, "Kernel_Name")
    per = collections.defaultdict(lambda: collections.defaultdict(list))
    ids = sorted({int(r["Dispatch_Id"]) for r in rows})
    order = {d: i for i, d in enumerate(ids)}
    for r in rows:
        per[order[int(r["Dispatch_Id"])] // 4][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for i, nq in enumerate(sizes):
        for c, v in sorted(per[i].items()):
            a = sum(v) / len(v)
            extra = f"  = {2 * a * 1024 / 1e9:.3f} GB fetched (x2 x1024: gfx950 correction)" if c == "FETCH_SIZE" else ""
            print(f"{nq:5d} queries  {c:28s} {a:18.1f} (n={len(v)}){extra}")
PY
