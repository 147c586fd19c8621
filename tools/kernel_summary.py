"""Cut rocprofv3 output of tools/kernel_workload.py into scenarios and write profiles/<tag>_kernels.json.

usage: python tools/kernel_summary.py <gpurun_out dir with manifest.json, trace/ and pmc_*/> <tag>
Per scenario: average duration of the named kernel over the timed reps (kernel trace), achieved GB/s or TFLOP/s from the
ALGORITHMIC bytes / flops in the manifest, fraction of the 8 TB/s HBM or 2.5 PFLOP/s dense fp16 MFMA peak, and - from the
separate --pmc passes - HBM-side traffic (2 x FETCH_SIZE x 1024, gfx950 correction), L2 hit rate, MFMA busy fraction."""
import collections, csv, glob, hashlib, json, os, sys

src, tag = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
from bench import csrc_hash, PEAK_FP16_MFMA_TFLOPS, PEAK_HBM_GBS  # noqa: E402

DELIM = "keys_to_result_kernel"
manifest = json.load(open(os.path.join(src, "manifest.json")))
csv.field_size_limit(1 << 30)


def segments(rows, name_key, order_key):
    """rows of one CSV -> list of per-scenario row lists (cut at every delimiter dispatch)."""
    rows = sorted(rows, key=lambda r: int(r[order_key]))
    segs, cur, seen = [], None, set()
    for r in rows:
        if DELIM in r[name_key]:
            did = r.get("Dispatch_Id")
            if did in seen:
                continue  # counter CSVs repeat a dispatch once per counter
            seen.add(did)
            if cur is not None:
                segs.append(cur)
            cur = []
        elif cur is not None:
            cur.append(r)
    return segs


sha = csrc_hash()
try:  # the hash of the build that ran (written by tools/kernel_workload.py next to the manifest)
    sha = open(os.path.join(src, "csrc_sha.txt")).read().strip() or sha
except Exception:
    pass
out = {"tag": tag, "csrc_sha": sha, "peaks": {"hbm_GBs": PEAK_HBM_GBS, "fp16_mfma_TFLOPs": PEAK_FP16_MFMA_TFLOPS},
       "scenarios": {}}
trace = glob.glob(os.path.join(src, "trace", "*kernel_trace.csv"))
if trace:
    segs = segments(list(csv.DictReader(open(trace[0]))), "Kernel_Name", "Dispatch_Id")
    assert len(segs) == len(manifest), (len(segs), len(manifest))
    for m, seg in zip(manifest, segs):
        mine = [r for r in seg if m["kernel"] in r["Kernel_Name"]]
        lpc = m["launches_per_call"]
        if len(mine) < (m["warm"] + m["reps"]) * lpc:  # the call was routed to another kernel than the manifest expected
            out["scenarios"][m["name"]] = {"kernel": m["kernel"], "skipped": f"only {len(mine)} matching dispatches",
                                           "kernels_seen": sorted({r["Kernel_Name"][:60] for r in seg})[:6]}
            continue
        timed = mine[m["warm"] * lpc:(m["warm"] + m["reps"]) * lpc]
        per_call_us = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in timed) / 1e3 / m["reps"]
        e = {"kernel": m["kernel"], "reps": m["reps"], "launches_per_call": lpc, "kernel_us_per_call": per_call_us,
             "bound": m["bound"], "note": m["note"], "vgpr": int(timed[0]["VGPR_Count"]), "lds_bytes": int(timed[0]["LDS_Block_Size"])}
        if m["bytes_per_call"]:
            e["algorithmic_bytes_per_call"] = m["bytes_per_call"]
            e["achieved_GBs"] = m["bytes_per_call"] / (per_call_us * 1e-6) / 1e9
            e["frac"] = e["achieved_GBs"] / PEAK_HBM_GBS
        if m["flops_per_call"]:
            e["algorithmic_flops_per_call"] = m["flops_per_call"]
            e["achieved_TFLOPs"] = m["flops_per_call"] / (per_call_us * 1e-6) / 1e12
            e["frac"] = e["achieved_TFLOPs"] / PEAK_FP16_MFMA_TFLOPS
        out["scenarios"][m["name"]] = e

for p in glob.glob(os.path.join(src, "pmc_*", "k_counter_collection.csv")):
    rows = list(csv.DictReader(open(p)))
    segs = segments(rows, "Kernel_Name", "Dispatch_Id")
    if len(segs) != len(manifest):
        print("skip", p, len(segs), len(manifest))
        continue
    for m, seg in zip(manifest, segs):
        lpc = m["launches_per_call"]
        by_counter = collections.defaultdict(list)
        for r in seg:
            if m["kernel"] in r["Kernel_Name"]:
                by_counter[r["Counter_Name"]].append(float(r["Counter_Value"]))
        e = out["scenarios"].setdefault(m["name"], {})
        if "skipped" in e:
            continue
        c = e.setdefault("counters_per_call", {})
        for name, vals in by_counter.items():
            vals = vals[m["warm"] * lpc:(m["warm"] + m["reps"]) * lpc]
            if vals:
                c[name] = sum(vals) / m["reps"]
for e in out["scenarios"].values():
    if "skipped" in e:
        continue
    c = e.get("counters_per_call", {})
    if "FETCH_SIZE" in c:
        e["hbm_read_bytes_per_call"] = 2.0 * c["FETCH_SIZE"] * 1024
    if "WRITE_SIZE" in c:
        e["hbm_write_bytes_per_call"] = c["WRITE_SIZE"] * 1024
    if "TCC_HIT_sum" in c and "TCC_MISS_sum" in c:
        e["l2_hit_rate"] = c["TCC_HIT_sum"] / max(1.0, c["TCC_HIT_sum"] + c["TCC_MISS_sum"])
    if "SQ_VALU_MFMA_BUSY_CYCLES" in c and "GRBM_GUI_ACTIVE" in c:
        e["mfma_busy_frac"] = (c["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024) / (c["GRBM_GUI_ACTIVE"] / 8)
    if "GRBM_GUI_ACTIVE" in c and "kernel_us_per_call" in e:
        e["effective_clock_GHz"] = (c["GRBM_GUI_ACTIVE"] / 8) / (e["kernel_us_per_call"] * 1e3)
    if "SQ_WAVE_CYCLES" in c:
        for n in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"):
            if n in c:
                e[n.lower() + "_frac"] = c[n] / c["SQ_WAVE_CYCLES"]
os.makedirs(os.path.join(root, "profiles"), exist_ok=True)
json.dump(out, open(os.path.join(root, "profiles", f"{tag}_kernels.json"), "w"), indent=1)
for n, e in out["scenarios"].items():
    print(f"{n:32s} {e.get('kernel_us_per_call', 0):10.1f} us  frac {e.get('frac', 0):.3f}  "
          f"{e.get('achieved_GBs', e.get('achieved_TFLOPs', 0)):.1f} {'GB/s' if 'achieved_GBs' in e else 'TFLOP/s'}")
