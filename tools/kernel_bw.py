"""Per-kernel bandwidth/throughput table for the secondary kernels from a rocprofv3 --kernel-trace CSV of
tools/kernel_bw_workload.py (known byte counts per launch).  usage: python tools/kernel_bw.py <kernel_trace.csv>"""
import csv, collections, sys, json
rows = list(csv.DictReader(open(sys.argv[1])))
dur = collections.defaultdict(list)
for r in rows:
    dur[r["Kernel_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)  # us
N, D = 4_000_000, 768
spec = {  # kernel substring -> (algorithmic bytes per launch, description)
    "pack_rows_vec_kernelIfLi1E": (N * D * 4 + N * D * 2 * 2 + N * 4, "pack 4M x 768 fp32 -> fp16 hi|lo (+norms)"),
    "pack_rows_vec_kernelIDF16_Li0E": (N * D * 2 + N * D * 2 + N * 4, "pack 4M x 768 fp16 -> fp16 (+norms)"),
    "gather_rows_kernel": (2 * 1_000_000 * D * 2, "gather 1M of 4M packed fp16 rows"),
    "km_reduce_kernel": (N * D * 2 + 1024 * D * 4, "k-means centroid sums, 4M x 768 fp16 rows, K = 1024"),
    "lvs_stream_kernel": (N * D * 2, "1 query x 4M x 768 fp16 (streaming search)"),
    "keys_to_result_kernel": (100_000 * 10 * (8 + 4 + 8), "decode 1M keys"),
}
out = {}
for name, ds in dur.items():
    for sub, (nbytes, desc) in spec.items():
        if sub in name:
            ds2 = sorted(ds)
            med = ds2[len(ds2) // 2]
            out[sub] = {"what": desc, "launches": len(ds), "median_us": med, "GB_per_s": nbytes / med / 1e3,
                        "frac_of_8TBs": nbytes / med / 1e3 / 8000}
print(json.dumps(out, indent=1))
