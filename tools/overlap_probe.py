"""Do the in-row-order centroid sums of one half of the rows run UNDER the assignment search of the other half (two streams,
no CU masks; development aid)?  10 M x 768 fp16 blob rows, K = 1 024."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import benchdata
from lotus_amd import _capi
from lotus_amd.backend import HipBackend

be = HipBackend("cuda:0")
n, d, K = 10_000_000, 768, 1024
xh, _ = benchdata.blobs(benchdata.CFG_KMEANS, n, d, K)
pk = be.pack(xh, _capi.PACK_F16)
del xh
cent = be.unpack(pk, be.to_device(np.arange(0, n, n // K)[:K]), raw=True)
cpk, cstats = be.kmeans_pack_centroids(cent, _capi.PACK_SPLIT)
h = n // 2 // 4096 * 4096
A, B = be.slice_rows(pk, 0, h), be.slice_rows(pk, h, n)
side = torch.cuda.Stream(device=be.device)
ws2 = torch.empty(int(be.lib.lvs_kmeans_accumulate_workspace_bytes(n, K)) + 256, dtype=torch.uint8, device=be.device)


def assign(x):
    return be.nearest(cpk, x, _capi.METRIC_L2, exact_scores=False, corpus_stats=cstats)


def sums_into(x, keys, sums, counts):
    be._c("lvs_kmeans_accumulate_keys", int(x.rows.data_ptr()), x.n, x.d, x.mode, int(keys.data_ptr()), 0, K, int(sums.data_ptr()),
          int(counts.data_ptr()), int(ws2.data_ptr()), int(ws2.numel()), be._stream())


def timed(fn, reps=3):
    fn(); be.synchronize()
    best = 1e9
    for _ in range(reps):
        be.synchronize(); t0 = time.perf_counter(); fn(); be.synchronize()
        best = min(best, time.perf_counter() - t0)
    return best * 1e3


keys_all = assign(pk)
kA = keys_all[:h].contiguous(); kB = keys_all[h:].contiguous()
sums = torch.zeros((K, d), dtype=torch.float32, device=be.device); counts = torch.zeros((K,), dtype=torch.float32, device=be.device)
print(f"assign all rows            {timed(lambda: assign(pk)):7.2f} ms", flush=True)
print(f"assign half A + half B     {timed(lambda: (assign(A), assign(B))):7.2f} ms", flush=True)
print(f"sums all rows              {timed(lambda: (sums.zero_(), counts.zero_(), sums_into(pk, keys_all, sums, counts))):7.2f} ms", flush=True)
ref = sums.clone()
def two():
    sums.zero_(); counts.zero_(); sums_into(A, kA, sums, counts); sums_into(B, kB, sums, counts)
print(f"sums half A then half B    {timed(two):7.2f} ms   bit-identical to one launch: {bool(torch.equal(sums, ref))}", flush=True)


def serial():
    assign(A); sums.zero_(); counts.zero_(); sums_into(A, kA, sums, counts); assign(B); sums_into(B, kB, sums, counts)


def overlapped():
    a = assign(A)
    ev = torch.cuda.current_stream(be.device).record_event()
    with torch.cuda.stream(side):
        side.wait_event(ev)
        sums.zero_(); counts.zero_(); sums_into(A, kA, sums, counts)
        done = side.record_event()
    b = assign(B)
    torch.cuda.current_stream(be.device).wait_event(done)
    sums_into(B, kB, sums, counts)


print(f"assign A, sums A, assign B, sums B   serial     {timed(serial):7.2f} ms", flush=True)
print(f"... sums A on a side stream under assign B      {timed(overlapped):7.2f} ms   sums identical: {bool(torch.equal(sums, ref))}", flush=True)


def parts_run(P, overlap):
    cut = [min(n, (n * i // P) // 4096 * 4096) for i in range(P)] + [n]
    xs = [be.slice_rows(pk, cut[i], cut[i + 1]) for i in range(P)]
    ks = [keys_all[cut[i]:cut[i + 1]].contiguous() for i in range(P)]
    main = torch.cuda.current_stream(be.device)

    def run():
        prev = None
        for i in range(P):
            assign(xs[i])
            ev = main.record_event()
            if overlap:
                with torch.cuda.stream(side):
                    side.wait_event(ev)
                    if i == 0:
                        sums.zero_(); counts.zero_()
                    sums_into(xs[i], ks[i], sums, counts)
                    prev = side.record_event()
            else:
                if i == 0:
                    sums.zero_(); counts.zero_()
                sums_into(xs[i], ks[i], sums, counts)
        if overlap:
            main.wait_event(prev)
    return run


for P in (2, 4, 8):
    a, b = timed(parts_run(P, False)), timed(parts_run(P, True))
    print(f"{P} parts: serial {a:7.2f} ms, sums on the side stream {b:7.2f} ms   sums identical: {bool(torch.equal(sums, ref))}", flush=True)
