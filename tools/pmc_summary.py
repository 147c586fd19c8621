"""Summarise rocprofv3 CSV output (kernel stats + separate --pmc passes) of `bench.py` into profiles/.

usage: python tools/pmc_summary.py <gpurun_out dir> <tag>    ->  profiles/<tag>_kernel_stats.csv, profiles/<tag>_pmc.json
HBM traffic follows /opt/skills/guides/MI355X_MICROARCH.md (section HBM): FETCH_SIZE / WRITE_SIZE are reported in KiB by
rocprofv3; on gfx950 FETCH_SIZE counts 64 B per 128-B request for wide (16 B/lane) streaming reads, so read bytes =
2 * FETCH_SIZE * 1024.  WRITE_SIZE is uncalibrated (taken as is); it is 5 orders of magnitude below the reads here."""
import collections, csv, glob, json, os, sys

src, tag = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
from bench import csrc_hash  # noqa: E402
out_dir = os.path.join(root, "profiles")
os.makedirs(out_dir, exist_ok=True)
KERNEL = "lvs_tile_kernel"
DOMINANT = None

stats = glob.glob(os.path.join(src, "prof_*", "*kernel_stats.csv"))
summary = {}
if stats:
    rows = list(csv.DictReader(open(stats[0])))
    keep = [r for r in rows if float(r["Percentage"]) >= 0.01 or KERNEL in r["Name"]]
    with open(os.path.join(out_dir, f"{tag}_kernel_stats.csv"), "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=rows[0].keys())
        w.writeheader()
        for r in keep:
            r = dict(r)
            if len(r["Name"]) > 160:
                r["Name"] = r["Name"][:157] + "..."
            w.writerow(r)
    # the dominant kernel of the search: the one of the library's search kernels with the largest total duration (r6: the
    # register-resident-queries kernel lvs_rj_kernel in the default configuration, one launch per chunk of <= 32 768 queries;
    # the list kernel lvs_tile_kernel<0, 4> before) - SEED-mode sample passes are other instantiations / far shorter
    cand = [r for r in rows if any(kn in r["Name"] for kn in ("lvs_rj_kernel", "lvs_rq_kernel", "lvs_tile_kernel"))]
    if cand:
        r = max(cand, key=lambda r: float(r["TotalDurationNs"]))
        DOMINANT = r["Name"]
        summary["kernel"] = r["Name"]
        summary["rocprof_kernel_avg_ms"] = float(r["AverageNs"]) / 1e6
        summary["rocprof_kernel_calls"] = int(r["Calls"])
        summary["rocprof_kernel_total_ms"] = float(r["TotalDurationNs"]) / 1e6

counters = collections.defaultdict(list)
for p in glob.glob(os.path.join(src, "pmc_*", "b_counter_collection.csv")):
    for r in csv.DictReader(open(p)):
        if (r["Kernel_Name"] == DOMINANT) if DOMINANT else (KERNEL in r["Kernel_Name"] and "<0, 4>" in r["Kernel_Name"]):
            counters[r["Counter_Name"]].append(float(r["Counter_Value"]))
            summary.setdefault("vgpr", int(r["VGPR_Count"]))
            summary.setdefault("sgpr", int(r["SGPR_Count"]))
            summary.setdefault("lds_bytes", int(r["LDS_Block_Size"]))
            summary.setdefault("grid", int(r["Grid_Size"]))
avg = {k: sum(v) / len(v) for k, v in counters.items()}
summary["counters_per_launch"] = avg
if "FETCH_SIZE" in avg:
    rd = 2.0 * avg["FETCH_SIZE"] * 1024
    wr = avg.get("WRITE_SIZE", 0.0) * 1024
    summary["hbm_read_bytes_per_launch"] = rd
    summary["hbm_write_bytes_per_launch"] = wr
    summary["traffic_bytes_per_launch"] = rd + wr
if "TCC_HIT_sum" in avg:
    summary["l2_hit_rate"] = avg["TCC_HIT_sum"] / (avg["TCC_HIT_sum"] + avg["TCC_MISS_sum"])
if "SQ_VALU_MFMA_BUSY_CYCLES" in avg and "GRBM_GUI_ACTIVE" in avg:
    # GRBM_GUI_ACTIVE is summed over the 8 XCDs; MFMA busy cycles over the 1024 SIMDs
    summary["mfma_busy_frac"] = (avg["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024) / (avg["GRBM_GUI_ACTIVE"] / 8)
if "SQ_LDS_IDX_ACTIVE" in avg and "GRBM_GUI_ACTIVE" in avg:
    # the LDS port's occupancy (rocprofv3's own LdsUtil expression: SQ_LDS_IDX_ACTIVE summed over the CUs / (busy cycles x CUs));
    # indexed operations only - the fragment reads (ds_read_b128) - the LDS-DMA writes of global_load_lds are not in it
    summary["lds_idx_active_frac"] = (avg["SQ_LDS_IDX_ACTIVE"] / 256) / (avg["GRBM_GUI_ACTIVE"] / 8)
    if "SQ_LDS_BANK_CONFLICT" in avg:
        summary["lds_bank_conflict_frac_of_active"] = avg["SQ_LDS_BANK_CONFLICT"] / avg["SQ_LDS_IDX_ACTIVE"]
if "SQ_WAVE_CYCLES" in avg:
    for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"):
        if c in avg:
            summary[c.lower() + "_frac"] = avg[c] / avg["SQ_WAVE_CYCLES"]
summary["tag"] = tag
# bench.py reports `traffic` only while the kernel sources still hash to the value of the build the counters were
# collected on: that is the hash the un-profiled bench run of the same call printed (gpurun_out/<tag>/bench.json)
sha = csrc_hash()
try:
    sha = json.load(open(os.path.join(src, "bench.json")))["roofline"]["csrc_sha"]
except Exception:
    pass
summary["csrc_sha"] = sha
json.dump(summary, open(os.path.join(out_dir, f"{tag}_pmc.json"), "w"), indent=1)
if "traffic_bytes_per_launch" in summary:
    json.dump(summary, open(os.path.join(out_dir, "latest_pmc.json"), "w"), indent=1)
print(json.dumps(summary, indent=1))
