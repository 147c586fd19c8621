#!/bin/bash
# usage (on the GPU box, from the repo root): tools/profile_kernels.sh <tag> [scenario substrings...]
# rocprofv3 kernel trace + separate --pmc passes (never combined with another trace domain) of tools/kernel_workload.py,
# summarised per scenario into profiles/<tag>_kernels.json by tools/kernel_summary.py.
tag=${1:-prof}; shift
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
out=gpurun_out/$tag; mkdir -p $out
W="python tools/kernel_workload.py $out $*"
timeout -k 5 400 rocprofv3 --kernel-trace --output-format csv -d $out/trace -o k -- $W > $out/trace.log 2>&1
for c in "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; do
  n=$(echo $c | cut -d" " -f1)
  timeout -k 5 400 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/pmc_$n -o k -- $W > $out/pmc_$n.log 2>&1
done
find $out -name "*.db" -delete
python tools/kernel_summary.py $out $tag 2>&1 | tail -30
