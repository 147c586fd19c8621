#!/bin/bash
# usage (GPU box, repo root): tools/r05_run.sh <tag> [pytest|bench|prof|kern ...]   - one gpurun call, everything under gpurun_out/<tag>/
tag=${1:-r05}; shift
what=${*:-pytest bench}
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
out=gpurun_out/$tag; mkdir -p $out
for w in $what; do
  case $w in
    pytest) timeout -k 10 1500 python -m pytest tests -m gpu -q --maxfail=25 --timeout 600 --durations=15 -p no:cacheprovider > $out/pytest_gpu.log 2>&1; tail -40 $out/pytest_gpu.log ;;
    opb) timeout -k 10 900 python tools/op_bench.py > $out/op_bench.json 2> $out/op_bench.err; tail -30 $out/op_bench.json ;;
    pypar) timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_random_parity.py -m gpu -q --maxfail=10 --timeout 600 --durations=8 -p no:cacheprovider > $out/pytest_par.log 2>&1; tail -25 $out/pytest_par.log ;;
    pykm) timeout -k 10 900 python -m pytest tests/test_gpu_kmeans.py -m gpu -q --timeout 600 -p no:cacheprovider > $out/pytest_km.log 2>&1; tail -15 $out/pytest_km.log ;;
    pyacc) timeout -k 10 600 python -m pytest tests/test_gpu_kmeans.py -m gpu -q -k "accumulate or counting or objective or golden or bounds" --timeout 600 -p no:cacheprovider > $out/pytest_acc.log 2>&1; tail -3 $out/pytest_acc.log ;;
    pydur) timeout -k 10 900 python -m pytest tests/test_gpu_dist.py tests/test_gpu_parity.py -m gpu -q -k "rccl or ranking or rank_all or 2d_split" --durations=12 --timeout 600 -p no:cacheprovider > $out/pytest_dur.log 2>&1; tail -25 $out/pytest_dur.log ;;
    bench) timeout -k 10 900 python bench.py > $out/bench.json 2> $out/bench.err; tail -c 1500 $out/bench.json; echo; tail -5 $out/bench.err ;;
    benchq) timeout -k 10 600 python bench.py --dedup-rows 0 $BENCHQ_ARGS --no-cpu-baseline > $out/bench_quick.json 2> $out/bench_quick.err; tail -c 1500 $out/bench_quick.json; echo; tail -5 $out/bench_quick.err ;;
    prof) SKIP_BENCH=1 timeout -k 10 1500 tools/profile_bench.sh $tag/prof > $out/prof.log 2>&1; tail -30 $out/prof.log ;;
    kern) timeout -k 10 1500 tools/profile_kernels.sh $tag/kern $KERN_SCEN > $out/kern.log 2>&1; tail -30 $out/kern.log ;;
    kmprof) timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/kmprof -o km -- python tools/kmeans_iter_workload.py > $out/kmprof.log 2>&1; tail -5 $out/kmprof.log ;;
    kmprofb) timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/kmprofb -o km -- python tools/kmeans_iter_workload.py 10000000 blobs > $out/kmprofb.log 2>&1; tail -5 $out/kmprofb.log ;;
    sweep) make -C lotus_amd/csrc tuning -j8 > $out/tuning_build.log 2>&1; timeout -k 10 600 python tools/small_batch_sweep.py > $out/small_batch_sweep.log 2>&1; cat $out/small_batch_sweep.log ;;
    sbtrace) timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/sbtrace -o sb -- python tools/small_batch_sweep.py trace > $out/sbtrace.log 2>&1; tail -3 $out/sbtrace.log ;;
    kmdebug) timeout -k 10 600 python tools/kmeans_bounds_debug.py > $out/kmdebug.log 2>&1; tail -60 $out/kmdebug.log ;;
    kmtime) timeout -k 10 600 python tools/kmeans_iter_workload.py 10000000 both blobs > $out/kmtime.log 2>&1; tail -20 $out/kmtime.log ;;
    kmtrace) timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/kmtrace -o km -- python tools/kmeans_iter_workload.py 10000000 bounds blobs > $out/kmtrace.log 2>&1; grep iterations $out/kmtrace.log ;;
    sweep5) timeout -k 10 900 python tools/r05_sweep.py $SWEEP > $out/sweep5.log 2>&1; grep -v "^\[" $out/sweep5.log | tail -80 ;;
    refgemm) timeout -k 10 300 python tools/ref_gemm.py > $out/ref_gemm.log 2>&1; cat $out/ref_gemm.log ;;
    fp32trace) timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/fp32trace -o f -- python tools/fp32_trace_workload.py > $out/fp32trace.log 2>&1; grep "per call" $out/fp32trace.log ;;
    fp32k1) timeout -k 10 300 python tools/fp32_trace_workload.py > $out/fp32k1.log 2>&1; grep "per call" $out/fp32k1.log ;;
    kmdpl) make -C lotus_amd/csrc tuning -j8 > $out/tuning_build.log 2>&1; timeout -k 10 600 python tools/km_reduce_probe.py dpl > $out/kmdpl.log 2>&1; grep "ms per call" $out/kmdpl.log ;;
    overlap) timeout -k 10 600 python tools/overlap_probe.py > $out/overlap.log 2>&1; grep " ms" $out/overlap.log ;;
    seedpool) make -C lotus_amd/csrc tuning -j8 > $out/tuning_build.log 2>&1; timeout -k 10 600 python tools/seed_pool_sweep.py > $out/seedpool.log 2>&1; grep "tiles per shard" $out/seedpool.log ;;
    kmparts) timeout -k 10 600 python tools/km_parts_probe.py > $out/kmparts.log 2>&1; grep -v "^\[" $out/kmparts.log | tail -20 ;;
    bench2) LOTUS_BENCH_REHEARSAL=1 timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 3 --warmup 1 > $out/bench2.json 2> $out/bench2.err; tail -c 1800 $out/bench2.json; echo; tail -5 $out/bench2.err ;;
    ab) for i in 1 2; do for l in lotus_amd/liblotus_hip.so lotus_amd/liblotus_hip_prevtile.so; do timeout -k 10 300 python tools/ab_probe.py $l 2>&1 | grep "fp16 100k" >> $out/ab.log; done; done; cat $out/ab.log ;;
    k12) timeout -k 10 200 python tools/fp32_k12_probe.py > $out/k12.log 2>&1; grep "per call" $out/k12.log ;;
    tcall) timeout -k 10 600 python tools/tcall_probe.py > $out/tcall.log 2>&1; cat $out/tcall.log ;;
    pyfix) timeout -k 10 900 python -m pytest tests -m gpu -q -k "$PYK" --timeout 600 --durations=12 -p no:cacheprovider > $out/pytest_fix.log 2>&1; tail -30 $out/pytest_fix.log ;;
    selfl) timeout -k 10 900 python -m pytest tests/test_gpu_dist.py -m gpu -q -k "self_launches" --timeout 900 -p no:cacheprovider > $out/pytest_selfl.log 2>&1; tail -15 $out/pytest_selfl.log ;;
    kmpipe) timeout -k 10 600 python tools/km_pipeline_probe.py > $out/kmpipe.log 2>&1; grep -v "^\[" $out/kmpipe.log | tail -8 ;;
    fp32stride) timeout -k 10 400 python tools/fp32_stride_probe.py > $out/fp32stride.log 2>&1; cat $out/fp32stride.log | tail -80 ;;
    kmskew) timeout -k 10 600 python tools/km_skew_probe.py > $out/kmskew.log 2>&1; grep -v "^\[" $out/kmskew.log | tail -30 ;;
    rq) timeout -k 10 600 python tools/rq_probe.py > $out/rq.log 2>&1; grep -v "^\[" $out/rq.log | tail -40 ;;
    rqg) timeout -k 10 900 python tools/rq_groups_probe.py > $out/rqg.log 2>&1; grep -v "^\[\|amdgpu.ids" $out/rqg.log | tail -50 ;;
    rqab) timeout -k 10 600 python tools/rq_ablate.py > $out/rqab.log 2>&1; grep -v "amdgpu.ids" $out/rqab.log | tail -30 ;;
    bench2self) LOTUS_BENCH_REHEARSAL=1 timeout -k 10 900 python bench.py --gpus 2 --steps 3 --warmup 1 > $out/bench2.json 2> $out/bench2.err; tail -c 2500 $out/bench2.json; echo; grep -v "amdgpu.ids\|OMP_NUM\|\*\*\*" $out/bench2.err | tail -5 ;;
    smoke) timeout -k 10 300 python -c 'import __graft_entry__ as g; g.smoke()' > $out/smoke.log 2>&1; tail -3 $out/smoke.log ;;
    *) echo "unknown step $w" ;;
  esac
done
find $out -name "*.db" -delete 2>/dev/null
echo done
