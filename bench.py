#!/usr/bin/env python
"""bench.py - sem_sim_join hot path on N MI355X GPUs of one node.

One "step" = one full pass of the hot path over one batch: every one of Q left rows (queries, d=768 fp16) is
searched against the N-row right index (corpus, d=768 fp16), k=10, exactly what `sem_sim_join` hands to
`VS.__call__` (lotus/sem_ops/sem_sim_join.py:134).  The corpus is row-sharded over the ranks, queries are replicated,
each rank runs the tiled MFMA distance + fused top-k kernel on its shard, then ONE RCCL all-gather of the per-shard
candidate keys (8 B each) and a merge + decode give every rank the final (D, I).  Inputs are resident in HBM before
the timed region; total work is fixed as N grows ("strong" scaling).

Prints ONE JSON line on rank 0 (see the contract in the task statement): value = Q * steps / time in queries/s.
Extra objects: "roofline" (dominant kernel vs the dense fp16 MFMA peak, timed with HIP events on the launch stream)
and, at N=1, "cpu_baseline" (the CPU oracle timed on a bounded sample of the same workload on this host) and
"recall_at_k" of the GPU result against that oracle sample.
"""
from __future__ import annotations

import argparse
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
    ap.add_argument("--cpu-sample", type=int, default=2048, help="queries in the CPU-oracle sample (N=1 only)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
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

    from lotus_amd import _capi
    from lotus_amd.backend import HipBackend

    be = HipBackend(device)
    n, d, nq, k = args.corpus, args.dim, args.queries, args.k
    xb, xq, planted = make_data(torch, device, n, d, nq)
    per = -(-n // world)
    lo, hi = min(n, rank * per), min(n, (rank + 1) * per)
    corpus = be.pack(xb[lo:hi], _capi.PACK_F16)  # this rank's shard, resident
    queries = be.pack(xq, _capi.PACK_F16)  # replicated
    if world > 1:
        del xb  # only the shard stays
    parts = torch.empty((world, nq, k), dtype=torch.int64, device=device) if world > 1 else None

    def step():
        keys = be.search_keys(corpus, queries, k, _capi.METRIC_IP, id_offset=lo)
        if world > 1:
            dist.all_gather_into_tensor(parts, keys)
            keys = be.merge_keys(parts)
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

    out = None
    if rank == 0:
        kernel_ms = ktot_ms / max(1, klaunches)
        flops_per_launch = 2.0 * nq * (hi - lo) * d  # 2*Q*N*d, N = rows of this rank's shard
        achieved = flops_per_launch / (kernel_ms * 1e-3) / 1e12
        alg_bytes = (hi - lo) * d * 2 + nq * d * 2 + nq * k * 12  # 8(d): every input once + outputs once
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
                                   f"corpus row-sharded over {world} GPU(s), RCCL all-gather top-k merge",
                       "queries": nq, "corpus_rows": n, "dim": d, "k": k, "shard_rows": hi - lo},
            "planted_neighbour_at_rank1": planted_at_1,
            "roofline": {"bound": "mfma", "achieved": achieved, "peak": PEAK_FP16_MFMA_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / PEAK_FP16_MFMA_TFLOPS, "traffic": pmc_traffic(n, nq, world),
                         "kernel": "lvs_tile_kernel<TOPK, 256x256>", "kernel_ms": kernel_ms, "launches": klaunches,
                         "algorithmic_flops_per_launch": flops_per_launch,
                         "algorithmic_bytes_per_launch": alg_bytes,
                         "hbm_frac_secondary": alg_bytes / (kernel_ms * 1e-3) / 1e9 / PEAK_HBM_GBS},
        }
        if world == 1 and not args.no_cpu_baseline:
            out.update(cpu_baseline(np, torch, xb, xq, D, I, args.cpu_sample, k))
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def pmc_traffic(n, nq, world):
    """HBM-side bytes per launch of the tile kernel from the committed rocprofv3 PMC passes of this same command
    (profiles/latest_pmc.json: FETCH_SIZE x 2 x 1024 per the gfx950 correction of MI355X_MICROARCH.md, + WRITE_SIZE x
    1024; separate --pmc runs, tools/pmc_summary.py).  Counters cannot be read from inside the timed process, so the
    figure is only reported for the configuration it was collected on (1 GPU, default sizes); otherwise null."""
    try:
        with open(os.path.join(ROOT, "profiles", "latest_pmc.json")) as f:
            pmc = json.load(f)
        if world == 1 and n == 1_000_000 and nq == 100_000:
            return pmc.get("traffic_bytes_per_launch")
    except Exception:
        pass
    return None


def cpu_baseline(np, torch, xb, xq, D, I, sample, k):
    """Time the CPU oracle (faiss-equivalent blocked sgemm + k-best collector, oracle/flat.py) on a bounded sample
    of the same workload - the first `sample` queries against the WHOLE corpus - and check the GPU result on it."""
    import oracle
    from oracle import cbind

    sample = min(sample, xq.shape[0])
    xb_h = xb.cpu().numpy().astype(np.float32)  # the same fp16 values, upcast
    xq_h = xq[:sample].cpu().numpy().astype(np.float32)
    threads = os.cpu_count() or 1
    try:
        from threadpoolctl import threadpool_info

        blas = [p.get("num_threads", 1) for p in threadpool_info() if p.get("user_api") == "blas"]
        if blas:
            threads = max(blas)
    except Exception:
        pass
    t0 = time.perf_counter()
    Dr, Ir = oracle.flat_search(xb_h, xq_h, k)
    dt = time.perf_counter() - t0
    Dg, Ig = D[:sample].cpu().numpy(), I[:sample].cpu().numpy()
    inter = sum(len(set(a.tolist()) & set(b.tolist())) for a, b in zip(Ir, Ig))
    # ids must match wherever the oracle's neighbouring scores are > 2e-5 apart (near-ties may swap)
    mism = Ir != Ig
    hard = 0
    for q, r in zip(*np.nonzero(mism)):
        gaps = [abs(float(Dr[q, r]) - float(Dr[q, r - 1]))] if r > 0 else []
        gaps.append(abs(float(Dr[q, r]) - float(Dr[q, r + 1])) if r + 1 < k else 0.0)
        hard += min(gaps) > 2e-5
    return {
        "cpu_baseline": {"value": sample / dt, "unit": "queries/s", "cores": int(threads), "kind": "port",
                         "sample": f"first {sample} queries x full {xb_h.shape[0]}-row corpus, d={xb_h.shape[1]}, k={k} "
                                   f"(oracle/flat.py: 4096x1024 sgemm blocks + C k-best collector), {dt:.1f} s",
                         "host_cpus": os.cpu_count(), "c_helper_threads": cbind.num_threads() if cbind.available() else 0},
        "recall_at_k": inter / float(Ir.size),
        "max_abs_score_err": float(np.abs(Dr - Dg).max()),
        "id_mismatches_outside_near_ties": int(hard),
    }


if __name__ == "__main__":
    main()
