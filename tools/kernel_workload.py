"""Exercise every kernel a roofline is claimed for, one scenario after another, for rocprofv3 (tools/profile_kernels.sh).

Scenarios are separated in the dispatch stream by ONE `keys_to_result_kernel` launch each (no scenario launches that
kernel itself), so tools/kernel_summary.py can cut the kernel trace / counter CSVs into scenarios without relying on
clocks.  The manifest (scenario name, kernel, warm-ups, reps, algorithmic bytes or flops per call) goes to
<out>/manifest.json.  usage: python tools/kernel_workload.py <out dir> [scenario substring ...]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from lotus_amd.backend import HipBackend
from lotus_amd import _capi

out_dir = sys.argv[1]
only = sys.argv[2:]
os.makedirs(out_dir, exist_ok=True)
be = HipBackend("cuda:0"); dev = be.device
be.PACK_CHUNK_ROWS = 1 << 30  # one launch per pack so that byte counts per launch are known
F16, SPLIT, IP, L2 = _capi.PACK_F16, _capi.PACK_SPLIT, _capi.METRIC_IP, _capi.METRIC_L2
g = torch.Generator(device=dev); g.manual_seed(1)

def gen(n, d, dtype=torch.float16):
    out = torch.empty((n, d), dtype=dtype, device=dev)
    for r0 in range(0, n, 1 << 18):
        r1 = min(n, r0 + (1 << 18))
        out[r0:r1] = torch.nn.functional.normalize(torch.randn((r1 - r0, d), generator=g, device=dev), dim=1).to(dtype)
    return out

N4, D = 4_000_000, 768
x16 = gen(N4, D)
p4m = be.pack(x16, F16)
p1m = be.slice_rows(p4m, 0, 1_000_000)
x384 = gen(2_000_000, 384)
p384 = be.pack(x384, F16)
j = torch.randint(0, 1_000_000, (100_000,), generator=g, device=dev)
xq = torch.nn.functional.normalize(0.7 * x16[j].float() + 0.7 * torch.nn.functional.normalize(
    torch.randn((100_000, D), generator=g, device=dev), dim=1), dim=1).to(torch.float16)
q100k = be.pack(xq, F16)
q384 = be.pack(x384[:32].contiguous(), F16)
x32 = x16[:1_000_000].float()
cent = gen(1024, D, torch.float32)
pc = be.pack(cent, SPLIT)            # fp32-accurate centroids (hi|lo): two K segments
pc16 = be.pack(cent.half(), F16)
pts = be.slice_rows(p4m, 0, 2_000_000)
assign = torch.randint(0, 1024, (N4,), generator=g, device=dev)
ids1m = torch.randperm(N4, generator=g, device=dev)[:1_000_000].contiguous()
dummy = torch.ones((1, 1), dtype=torch.int64, device=dev)
be.synchronize()

manifest = []
def scenario(name, kernel, fn, reps, warm=2, bound="hbm", bytes_per_call=None, flops_per_call=None, launches_per_call=1, note=""):
    if only and not any(s in name for s in only):
        return
    be.keys_to_result(dummy, IP)  # delimiter
    for _ in range(warm + reps):
        fn()
    be.synchronize()
    manifest.append(dict(name=name, kernel=kernel, warm=warm, reps=reps, bound=bound, bytes_per_call=bytes_per_call,
                         flops_per_call=flops_per_call, launches_per_call=launches_per_call, note=note))

def ld(p): return int(p.rows.shape[1])
# ---- HBM-bound: the small-batch streaming kernel (the literal sem_search) ----
MAIN = ", false>(LvsStreamArgs)"  # the scan itself (the SEED-mode launch over the sample carries ", true>")
scenario("stream_1q_x_1M_d768", MAIN, lambda: be.search_keys(p1m, be.slice_rows(q100k, 0, 1), 10, IP), 10,
         bytes_per_call=1_000_000 * ld(p1m) * 2)
for nq_s, note_s in ((32, "one 32-query block per workgroup, thresholds seeded from 32 768 sample rows"),
                     (64, "two blocks per workgroup (8 waves)"), (96, "three blocks per workgroup")):
    scenario(f"stream_{nq_s}q_x_1M_d768", MAIN, lambda n_=nq_s: be.search_keys(p1m, be.slice_rows(q100k, 0, n_), 10, IP), 10,
             bytes_per_call=1_000_000 * ld(p1m) * 2, note=note_s)
# beyond one sibling group (97 .. 256 queries) and up to 64 query tiles: the list kernel with thresholds seeded from a sample
for nq_s, kern_s, note_s in ((128, "lvs_tile_kernel<0, 2>", "one 128-query tile x 245 slabs, seeded"),
                             (192, "lvs_tile_kernel<0, 4>", "one 256-query tile (three quarters full) x 245 slabs, seeded"),
                             (256, "lvs_tile_kernel<0, 4>", "one 256-query tile x 245 slabs, seeded")):
    scenario(f"list_{nq_s}q_x_1M_d768", kern_s, lambda n_=nq_s: be.search_keys(p1m, be.slice_rows(q100k, 0, n_), 10, IP), 10,
             bytes_per_call=1_000_000 * ld(p1m) * 2, note=note_s)
scenario("list_1024q_x_1M_d768", "lvs_tile_kernel<0, 4>", lambda: be.search_keys(p1m, be.slice_rows(q100k, 0, 1024), 10, IP), 5,
         bound="mfma", flops_per_call=2.0 * 1024 * 1_000_000 * D, note="four query tiles x 127 slabs, seeded")
scenario("stream_1q_x_4M_d768", MAIN, lambda: be.search_keys(p4m, be.slice_rows(q100k, 0, 1), 10, IP), 10,
         bytes_per_call=N4 * ld(p4m) * 2)
scenario("stream_32q_x_4M_d768", MAIN, lambda: be.search_keys(p4m, be.slice_rows(q100k, 0, 32), 10, IP), 10,
         bytes_per_call=N4 * ld(p4m) * 2)
scenario("stream_1q_x_2M_d384", MAIN, lambda: be.search_keys(p384, be.slice_rows(q384, 0, 1), 10, IP), 10,
         bytes_per_call=2_000_000 * ld(p384) * 2, note="d = 384 = BASELINE configs[0]'s dimension")
scenario("stream_32q_x_2M_d384", MAIN, lambda: be.search_keys(p384, q384, 10, IP), 10,
         bytes_per_call=2_000_000 * ld(p384) * 2)
# ---- MFMA-bound: the tile kernel in its modes ----
shard = be.slice_rows(p4m, 0, 125_000)
scenario("topk_100k_x_125k_shard", "lvs_tile_kernel<0, 4>", lambda: be.search_keys(shard, q100k, 10, IP), 5, bound="mfma",
         flops_per_call=2.0 * 100_000 * 125_000 * D, note="8-GPU shard shape of BASELINE configs[2]")
scenario("topk_12500_x_1M_split8x1", "lvs_tile_kernel<0, 4>", lambda: be.search_keys(p1m, be.slice_rows(q100k, 0, 12_500), 10, IP), 5,
         bound="mfma", flops_per_call=2.0 * 12_500 * 1_000_000 * D, note="per-GPU shape of the 8 x 1 query split: 49 query tiles")
scenario("topk_10k_x_1M_cfg2", "lvs_tile_kernel<0, 4>", lambda: be.search_keys(p1m, be.slice_rows(q100k, 0, 10_000), 10, IP), 5,
         bound="mfma", flops_per_call=2.0 * 10_000 * 1_000_000 * D, note="BASELINE configs[1]")
scenario("topk_100k_x_1M_cfg3", "lvs_tile_kernel<0, 4>", lambda: be.search_keys(p1m, q100k, 10, IP), 3, warm=1,
         bound="mfma", flops_per_call=2.0 * 100_000 * 1_000_000 * D, note="BASELINE configs[2] on one GPU (bench.py's step)")
NR = 1_000_000
nchunks = -(-NR // be.RANGE_CHUNK_ROWS)
scenario("range_selfjoin_1M", "lvs_tile_kernel<3, 4>", lambda: be.range_join(p1m, p1m, 0.95, IP, q_row0=0), 1, warm=1,
         bound="mfma", flops_per_call=1.0 * NR * NR * D, launches_per_call=nchunks,
         note="sem_dedup threshold self-join (cfg4 shape at 1 M rows); algorithmic flops = N^2 d (each unordered pair once)")
scenario("top1_kmeans_2M_x_1024_hilo", "lvs_tile_kernel<2, 4>", lambda: be.search_keys(pc, pts, 1, L2, one_pass=False), 3, warm=1,
         bound="mfma", flops_per_call=2.0 * 2_000_000 * 1024 * D,
         note="k-means assignment, fp16 points x fp32-accurate (hi|lo) centroids: 2 K segments = 2x the MFMA work")
scenario("top1_kmeans_2M_x_1024_fp16c", "lvs_tile_kernel<2, 4>", lambda: be.search_keys(pc16, pts, 1, L2), 3, warm=1,
         bound="mfma", flops_per_call=2.0 * 2_000_000 * 1024 * D, note="same with fp16 centroids (1 K segment)")
scenario("nearest3_2M_x_1024", "lvs_assign_kernel", lambda: be.nearest(pc, pts, L2), 3, warm=1,
         bound="mfma", flops_per_call=2.0 * 2_000_000 * 1024 * D,
         note="certified k-means assignment: ONE pass over the hi parts + margin certificate (same winners as the hi|lo search)")
x32m = x16[:1_000_000].float()
x32m += 1e-4 * torch.randn(x32m.shape, generator=g, device=dev)
c32 = be.pack(x32m, SPLIT)
q32 = be.pack(xq[:10_000].float() + 1e-4 * torch.randn((10_000, D), generator=g, device=dev), SPLIT)
del x32m
scenario("fp32_topk_10k_x_1M_plain", "lvs_tile_kernel<0, 4>", lambda: be.search_keys(c32, q32, 10, IP, one_pass=False), 3, warm=1,
         bound="mfma", flops_per_call=2.0 * 10_000 * 1_000_000 * D, note="fp32 embeddings (hi|lo rows), plain search: three K segments")
scenario("fp32_topk_10k_x_1M_one_pass", "lvs_tile_kernel<0, 4>", lambda: be.search_keys(c32, q32, 10, IP, one_pass=True), 3, warm=1,
         bound="mfma", flops_per_call=2.0 * 10_000 * 1_000_000 * D, launches_per_call=1,
         note="same result from one pass over the hi parts with 15 list slots (+ rescoring, certificate, ~1 % of the queries searched again: those launches are not in this figure)")
del c32, q32
# ---- HBM-bound helpers ----
scenario("km_reduce_4M_x_1024", "km_reduce_kernel", lambda: be.kmeans_accumulate(p4m, assign, 1024), 3, warm=1,
         bytes_per_call=N4 * ld(p4m) * 2)
# (rocprofv3 leaves kernels with _Float16 template arguments mangled: match the mangled fragment)
scenario("pack_f16_4M", "pack_rows_vec_kernelIDF16_Li0", lambda: be.pack(x16, F16), 3, warm=1,
         bytes_per_call=N4 * (D * 2 + ld(p4m) * 2 + 4))
scenario("pack_f32_hilo_1M", "pack_rows_vec_kernelIfLi1", lambda: be.pack(x32, SPLIT), 3, warm=1,
         bytes_per_call=1_000_000 * (D * 4 + 2 * ld(p4m) * 2 + 4))
scenario("gather_1M_rows", "gather_rows_kernel", lambda: be.gather(p4m, ids1m), 3, warm=1,
         bytes_per_call=1_000_000 * 2 * ld(p4m) * 2)
be.keys_to_result(dummy, IP)
be.synchronize()
json.dump(manifest, open(os.path.join(out_dir, "manifest.json"), "w"), indent=1)
from bench import csrc_hash  # noqa: E402
open(os.path.join(out_dir, "csrc_sha.txt"), "w").write(csrc_hash())
print("scenarios:", [m["name"] for m in manifest])
