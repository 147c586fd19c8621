#!/usr/bin/env python
"""bench.py - sem_sim_join hot path on N MI355X GPUs of one node.

One "step" = one full pass of the hot path over one batch: every one of Q left rows (queries, d=768 fp16) is
searched against the N-row right index (corpus, d=768 fp16), k=10, exactly what `sem_sim_join` hands to
`VS.__call__` (lotus/sem_ops/sem_sim_join.py:134).  The corpus is row-sharded over the ranks, queries are replicated,
each rank runs the tiled MFMA distance + fused top-k kernel on its shard, then ONE RCCL all-gather of the per-shard
candidate keys (8 B each) and a merge + decode give every rank the final (D, I).  Inputs are resident in HBM before
the timed region; total work is fixed as N grows ("strong" scaling).

Prints ONE JSON line on rank 0 (see the contract in the task statement): value = Q * steps / time in queries/s.
Extra objects:
  "roofline"      dominant kernel vs the dense fp16 MFMA peak, timed with HIP events on the launch stream;
                  `traffic` = HBM-side bytes per launch from the committed rocprofv3 PMC passes, reported only when the
                  kernel sources are byte-identical to the ones the counters were collected on (else null);
  "cpu_baseline"  (N=1) the CPU comparator (oracle/blas_twin.py: faiss's BLAS path on torch-CPU, all host cores) timed on a
                  bounded sample of the same workload;
  "recall_at_k" / "id_mismatches_outside_near_ties" / "max_abs_score_err": the GPU result of the LAST timed step checked
                  against the CPU oracle (oracle/flat.py) on a query sample - at every N, rank 0;
  "legs"          (N=1) short secondary measurements in the same process: BASELINE configs[1] (10k x 1M), the literal
                  single-query sem_search (HBM-bound streaming kernel), the per-GPU shard shapes at N = 8 / 4 / 2 (100k x 125k / 250k / 500k) and T_call
                  (`HipVS.__call__` host ndarray -> host (D, I), PCIe included), the per-GPU shape of the query split and the fp32-embeddings
                  call (plain vs certified one-pass) - each with kernel ms and roofline fraction.
"""
from __future__ import annotations

import argparse
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP16_MFMA_TFLOPS = 2500.0  # dense, /opt/skills/guides/MI355X_MICROARCH.md
PEAK_HBM_GBS = 8000.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--queries", type=int, default=100_000)
    ap.add_argument("--corpus", type=int, default=1_000_000)
    ap.add_argument("--dim", type=int, default=768)
    ap.add_argument("--k", type=int, default=10)
    ap.add_argument("--cpu-sample", type=int, default=8192, help="queries timed on the CPU comparator (N=1 only)")
    ap.add_argument("--check-sample", type=int, default=512, help="queries re-checked against the CPU oracle (rank 0)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-legs", action="store_true", help="skip the secondary legs (N=1)")
    return ap.parse_args()


def make_data(torch, device, n, d, nq):
    """Synthetic embeddings (BASELINE.md section 2 recipe, generated on the GPU): unit-norm Gaussian corpus rounded
    to fp16; queries = normalize(0.7 x[j] + 0.7 u) -> a planted neighbour at cos ~ 0.71.  Same seed on every rank."""
    g = torch.Generator(device=device)
    g.manual_seed(20260923)
    xb = torch.empty((n, d), dtype=torch.float16, device=device)
    blk = 262144
    for r0 in range(0, n, blk):
        r1 = min(n, r0 + blk)
        x = torch.randn((r1 - r0, d), generator=g, device=device, dtype=torch.float32)
        xb[r0:r1] = torch.nn.functional.normalize(x, dim=1).to(torch.float16)
    j = torch.randint(0, n, (nq,), generator=g, device=device)
    u = torch.nn.functional.normalize(torch.randn((nq, d), generator=g, device=device, dtype=torch.float32), dim=1)
    xq = torch.nn.functional.normalize(0.7 * xb[j].float() + 0.7 * u, dim=1).to(torch.float16)
    return xb, xq, j


def csrc_hash() -> str:
    """sha256 over the kernel sources + the C-ABI header: ties committed counter data to the binary that ran."""
    h = hashlib.sha256()
    base = os.path.join(ROOT, "lotus_amd", "csrc")
    names = sorted(f for f in os.listdir(base) if f.endswith((".hip", ".h")))
    for p in [os.path.join(base, f) for f in names] + [os.path.join(ROOT, "include", "lotus_hip.h")]:
        with open(p, "rb") as f:
            h.update(os.path.basename(p).encode() + b"\0" + f.read())
    return h.hexdigest()[:16]


def main():
    args = parse()
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC only on this driver; before any HIP call
    import numpy as np
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with: python -m torch.distributed.run --nnodes=1 --nproc-per-node N "
                             "--master-addr 127.0.0.1 --master-port P bench.py --gpus N ...")
        args.gpus = world
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        dist.init_process_group("nccl", device_id=device)  # RCCL

    from lotus_amd import _capi, _dist
    from lotus_amd.backend import HipBackend

    be = HipBackend(device)
    n, d, nq, k = args.corpus, args.dim, args.queries, args.k
    xb, xq, planted = make_data(torch, device, n, d, nq)
    per = -(-n // world)
    lo, hi = min(n, rank * per), min(n, (rank + 1) * per)
    corpus = be.pack(xb[lo:hi], _capi.PACK_F16)  # this rank's shard, resident
    queries = be.pack(xq, _capi.PACK_F16)  # replicated
    if world > 1 and rank != 0:
        del xb  # only the shard stays (rank 0 keeps the rows for the oracle check after the timed region)

    def step():
        keys = be.search_keys(corpus, queries, k, _capi.METRIC_IP, id_offset=lo)
        if world > 1:
            keys = be.merge_keys(_dist.all_gather_rows(keys))  # one RCCL all-gather of [Q,k] uint64 keys + merge
        return be.keys_to_result(keys, _capi.METRIC_IP)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(device)

    for _ in range(args.warmup):
        D, I = step()
    barrier()
    be.timing_enable(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        D, I = step()
    barrier()
    dt = time.perf_counter() - t0
    ktot_ms, klaunches = be.timing_read()
    be.timing_enable(False)
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # ---- sanity on the result (outside the timed region) ----
    planted_at_1 = float((I[:, 0] == planted).float().mean().item())

    if rank == 0:
        kernel_ms = ktot_ms / max(1, klaunches)
        flops_per_launch = 2.0 * nq * (hi - lo) * d  # 2*Q*N*d, N = rows of this rank's shard
        achieved = flops_per_launch / (kernel_ms * 1e-3) / 1e12
        alg_bytes = (hi - lo) * d * 2 + nq * d * 2 + nq * k * 12  # 8(d): every input once + outputs once
        traffic, traffic_src = pmc_traffic(n, nq, world)
        out = {
            "metric": "sem_sim_join queries/sec (d=768, k=10, exact top-k, recall vs CPU oracle)",
            "value": nq * args.steps / dt,
            "unit": "queries/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f16",
            "data": "synthetic",
            "config": {"workload": f"sem_sim_join: {nq} left x {n} right rows, d={d} fp16, k={k}, inner product; "
                                   f"corpus row-sharded over {world} GPU(s), RCCL all-gather top-k merge; "
                                   "timed device-resident queries -> device-resident (D, I)",
                       "queries": nq, "corpus_rows": n, "dim": d, "k": k, "shard_rows": hi - lo},
            "planted_neighbour_at_rank1": planted_at_1,
            "roofline": {"bound": "mfma", "achieved": achieved, "peak": PEAK_FP16_MFMA_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / PEAK_FP16_MFMA_TFLOPS, "traffic": traffic, "traffic_source": traffic_src,
                         "kernel": "lvs_tile_kernel<TOPK, 256x256>", "kernel_ms": kernel_ms, "launches": klaunches,
                         "algorithmic_flops_per_launch": flops_per_launch,
                         "algorithmic_bytes_per_launch": alg_bytes,
                         "hbm_frac_secondary": alg_bytes / (kernel_ms * 1e-3) / 1e9 / PEAK_HBM_GBS,
                         "csrc_sha": csrc_hash()},
        }
        if args.check_sample > 0:
            out.update(oracle_check(np, xb, xq, D, I, args.check_sample, k))
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(np, xb, xq, args.cpu_sample, k)
        if world == 1 and not args.no_legs:
            out["legs"] = secondary_legs(np, torch, be, _capi, xb, xq, corpus, queries, k)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def pmc_traffic(n, nq, world):
    """HBM-side bytes per launch of the tile kernel from the committed rocprofv3 PMC passes of this same command
    (profiles/latest_pmc.json: FETCH_SIZE x 2 x 1024 per the gfx950 correction of MI355X_MICROARCH.md, + WRITE_SIZE x
    1024; separate --pmc runs, tools/pmc_summary.py).  Counters cannot be read from inside the timed process, so the
    figure is a committed measurement - and it is only reported when (i) the configuration is the one it was
    collected on (1 GPU, default sizes) and (ii) the kernel sources hash to the value stamped at collection time.
    -> (bytes or None, provenance string)."""
    try:
        with open(os.path.join(ROOT, "profiles", "latest_pmc.json")) as f:
            pmc = json.load(f)
    except Exception:
        return None, "no committed PMC summary"
    if not (world == 1 and n == 1_000_000 and nq == 100_000):
        return None, "PMC passes exist for the 1-GPU default configuration only"
    have, want = pmc.get("csrc_sha"), csrc_hash()
    if have != want:
        return None, f"profiles/latest_pmc.json was collected on csrc {have}, this build is {want}: stale, not reported"
    return pmc.get("traffic_bytes_per_launch"), f"profiles/latest_pmc.json ({pmc.get('tag', '?')}, csrc {have})"


def _near_tie_mismatches(np, Dr, Ir, Ig, k):
    """ids must match wherever the oracle's neighbouring scores are > 2e-5 apart (near-ties may swap)."""
    hard = 0
    for q, r in zip(*np.nonzero(Ir != Ig)):
        gaps = [abs(float(Dr[q, r]) - float(Dr[q, r - 1]))] if r > 0 else []
        gaps.append(abs(float(Dr[q, r]) - float(Dr[q, r + 1])) if r + 1 < k else 0.0)
        hard += min(gaps) > 2e-5
    return int(hard)


def oracle_check(np, xb, xq, D, I, sample, k):
    """The GPU result of the last timed step against the CPU oracle (oracle/flat.py: 4096 x 1024 sgemm blocks + k-best
    collector with faiss's tie rule) on the first `sample` queries x the WHOLE corpus.  Runs at every N on rank 0."""
    import oracle

    sample = min(sample, xq.shape[0])
    xb_h = xb.cpu().numpy().astype(np.float32)  # the same fp16 values, upcast
    xq_h = xq[:sample].cpu().numpy().astype(np.float32)
    Dr, Ir = oracle.flat_search(xb_h, xq_h, k)
    Dg, Ig = D[:sample].cpu().numpy(), I[:sample].cpu().numpy()
    inter = sum(len(set(a.tolist()) & set(b.tolist())) for a, b in zip(Ir, Ig))
    return {"recall_at_k": inter / float(Ir.size), "max_abs_score_err": float(np.abs(Dr - Dg).max()),
            "id_mismatches_outside_near_ties": _near_tie_mismatches(np, Dr, Ir, Ig, k), "oracle_check_queries": sample}


def cpu_baseline(np, xb, xq, sample, k):
    """Time the CPU comparator on a bounded sample of the same workload - the first `sample` queries against the WHOLE
    corpus, all host cores.  The comparator is faiss's BLAS search path (blocked sgemm + k-best collector) as one fused
    C + OpenMP loop nest with an AVX-512 micro-kernel (oracle/c/lvs_blas_twin.c); if that library is missing the
    torch-CPU version (MKL sgemm + topk) is timed instead.  kind = "port": real faiss-cpu is not installable here."""
    from oracle import blas_twin

    sample = min(sample, xq.shape[0])
    xb_h = xb.cpu().numpy().astype(np.float32)
    xq_h = xq[:sample].cpu().numpy().astype(np.float32)
    impl = "oracle/c/lvs_blas_twin.c (C + OpenMP, AVX-512 12x32 sgemm micro-kernel fused with the k-best collector)"
    fn = blas_twin.flat_search_c
    if not blas_twin.c_available():
        impl = f"oracle/blas_twin.py (torch-CPU: {blas_twin.QUERY_BLOCK} x {blas_twin.DB_BLOCK} MKL sgemm blocks + topk)"
        fn = blas_twin.flat_search_blas
    fn(xb_h[:65536], xq_h[:256], k)  # thread pool / page warm-up, not timed
    t0 = time.perf_counter()
    _, _, threads = fn(xb_h, xq_h, k)
    dt = time.perf_counter() - t0
    flops = 2.0 * sample * xb_h.shape[0] * xb_h.shape[1]
    return {"value": sample / dt, "unit": "queries/s", "cores": int(threads), "kind": "port",
            "sample": f"first {sample} queries x full {xb_h.shape[0]}-row corpus, d={xb_h.shape[1]}, k={k}; {impl}; {dt:.1f} s",
            "gflops": flops / dt / 1e9, "host_cpus": os.cpu_count()}


def secondary_legs(np, torch, be, _capi, xb, xq, corpus, queries, k):
    """Short measurements of the other regimes of the same path, same process, same resident data (N=1)."""
    legs = {}
    d = int(xb.shape[1])

    def kernel_leg(cb, cq, reps, kk=k):
        for _ in range(2):
            be.search_keys(cb, cq, kk, _capi.METRIC_IP)
        be.synchronize()
        be.timing_enable(True)
        t0 = time.perf_counter()
        for _ in range(reps):
            keys = be.search_keys(cb, cq, kk, _capi.METRIC_IP)
            be.keys_to_result(keys, _capi.METRIC_IP)
        be.synchronize()
        wall = (time.perf_counter() - t0) / reps
        tot, cnt = be.timing_read()
        be.timing_enable(False)
        return tot / max(cnt, 1), wall * 1e3

    # BASELINE configs[1]: 10k queries x 1M rows, MFMA-bound
    q10k = be.slice_rows(queries, 0, min(10_000, queries.n))
    kms, wms = kernel_leg(corpus, q10k, 5)
    fl = 2.0 * q10k.n * corpus.n * d
    legs["cfg2_10k_x_1M"] = {"kernel_ms": kms, "ms_per_call": wms, "queries_per_s": q10k.n / (wms * 1e-3), "bound": "mfma",
                             "achieved_tflops": fl / (kms * 1e-3) / 1e12, "frac": fl / (kms * 1e-3) / 1e12 / PEAK_FP16_MFMA_TFLOPS}
    # the literal sem_search: ONE query per call (sem_search.py:121-122) -> lvs_stream_kernel, HBM-bound
    q1 = be.slice_rows(queries, 0, 1)
    kms, wms = kernel_leg(corpus, q1, 20)
    by = corpus.n * int(corpus.rows.shape[1]) * 2.0  # every corpus byte exactly once
    legs["sem_search_1_x_1M"] = {"kernel_ms": kms, "ms_per_call": wms, "bound": "hbm", "kernel": "lvs_stream_kernel",
                                 "achieved_gbs": by / (kms * 1e-3) / 1e9, "frac": by / (kms * 1e-3) / 1e9 / PEAK_HBM_GBS,
                                 "algorithmic_bytes_per_launch": by}
    q32 = be.slice_rows(queries, 0, 32)
    kms, wms = kernel_leg(corpus, q32, 20)
    legs["stream_32_x_1M"] = {"kernel_ms": kms, "ms_per_call": wms, "bound": "hbm", "kernel": "lvs_stream_kernel",
                              "achieved_gbs": by / (kms * 1e-3) / 1e9, "frac": by / (kms * 1e-3) / 1e9 / PEAK_HBM_GBS}
    # the 8-GPU shard shape of BASELINE configs[2]: 100k queries x 125k rows per GPU
    shard = be.slice_rows(corpus, 0, min(corpus.n, 125_000))
    kms, wms = kernel_leg(shard, queries, 5)
    fl = 2.0 * queries.n * shard.n * d
    legs["shard_100k_x_125k"] = {"kernel_ms": kms, "ms_per_call": wms, "bound": "mfma",
                                 "achieved_tflops": fl / (kms * 1e-3) / 1e12,
                                 "frac": fl / (kms * 1e-3) / 1e12 / PEAK_FP16_MFMA_TFLOPS,
                                 "node_queries_per_s_if_8_gpus": queries.n / (kms * 1e-3)}
    # the per-GPU shapes of the same join at N = 2 and N = 4 (500 k / 250 k rows of the corpus per GPU)
    for rows, ng in ((500_000, 2), (250_000, 4)):
        sh = be.slice_rows(corpus, 0, min(corpus.n, rows))
        kms, wms = kernel_leg(sh, queries, 3)
        fl = 2.0 * queries.n * sh.n * d
        legs[f"shard_100k_x_{rows // 1000}k"] = {"kernel_ms": kms, "ms_per_call": wms, "bound": "mfma",
                                                  "achieved_tflops": fl / (kms * 1e-3) / 1e12,
                                                  "frac": fl / (kms * 1e-3) / 1e12 / PEAK_FP16_MFMA_TFLOPS,
                                                  f"node_queries_per_s_if_{ng}_gpus": queries.n / (kms * 1e-3)}
    # ... and what each of 8 GPUs does under the QUERY split of the same join (HipVS(shard="queries"): corpus replicated,
    # 12 500 queries per GPU against all 1 M rows, finished lists all-gathered, no merge)
    q8 = be.slice_rows(queries, 0, min(queries.n, 12_500))
    kms, wms = kernel_leg(corpus, q8, 5)
    fl = 2.0 * q8.n * corpus.n * d
    legs["query_split_12500_x_1M"] = {"kernel_ms": kms, "ms_per_call": wms, "bound": "mfma",
                                      "achieved_tflops": fl / (kms * 1e-3) / 1e12,
                                      "frac": fl / (kms * 1e-3) / 1e12 / PEAK_FP16_MFMA_TFLOPS,
                                      "node_queries_per_s_if_8_gpus": 8 * q8.n / (kms * 1e-3)}
    # LOTUS's default storage: fp32 embeddings (fp16 hi|lo pairs on the device).  10k queries x the same 1M rows, plain
    # search (three K segments) vs the certified one-pass search (same exact result)
    xb32 = torch.nn.functional.normalize(xb.float() + 1e-4 * torch.randn_like(xb, dtype=torch.float32), dim=1)
    c32 = be.pack(xb32, _capi.PACK_SPLIT)
    del xb32
    q32 = be.pack(torch.nn.functional.normalize(xq[:10_000].float() + 1e-4 * torch.randn((10_000, d), device=xq.device), dim=1),
                  _capi.PACK_SPLIT)
    res32 = {}
    for tag, one_pass in (("plain_3_segments", False), ("one_pass_certified", True)):
        stats = {}
        for _ in range(2):
            be.search_keys(c32, q32, k, _capi.METRIC_IP, one_pass=one_pass)
        be.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            be.search_keys(c32, q32, k, _capi.METRIC_IP, one_pass=one_pass, stats=stats)
        be.synchronize()
        ms = (time.perf_counter() - t0) / 3 * 1e3
        res32[tag] = {"ms_per_call": ms, "algorithmic_tflops": 2.0 * q32.n * c32.n * d / (ms * 1e-3) / 1e12}
        if one_pass:
            res32[tag]["uncertified_fraction"] = stats["uncertified"] / max(1, stats["queries"])
    legs["fp32_embeddings_10k_x_1M"] = res32
    del c32, q32
    # T_call: VS.__call__(host ndarray) -> host (D, I), corpus resident; includes packing the queries,
    # the H2D copy of 154 MB and the D2H copy of the results
    from lotus_amd.vs import HipVS, _Resident

    vs = HipVS(backend=be, storage="fp16")
    vs._resident["bench"] = _Resident(vecs=None, packed=corpus, n=corpus.n, d=d, lo=0, hi=corpus.n)
    vs.index_dir = "bench"
    xq_h = xq.cpu().numpy()
    vs(xq_h[:1000], k)
    ts = []
    for _ in range(3):
        t0 = time.perf_counter()
        out = vs(xq_h, k)
        ts.append(time.perf_counter() - t0)
    tcall = sorted(ts)[1]
    assert out.indices.shape == (xq_h.shape[0], k)
    legs["t_call_host_to_host"] = {"ms_per_call": tcall * 1e3, "queries_per_s": xq_h.shape[0] / tcall,
                                   "note": "HipVS.__call__(numpy fp16 [Q,d]) -> numpy (D, I); pageable host memory, PCIe included"}
    return legs


if __name__ == "__main__":
    main()
