#!/bin/bash
# usage (on the GPU box, from the repo root): tools/profile_bench.sh <tag>
# Collects, for `python bench.py` at the default (judged) configuration:
#   gpurun_out/<tag>/bench.json                      the bench line itself (un-profiled run)
#   gpurun_out/<tag>/prof_stats/*kernel_stats.csv    rocprofv3 --kernel-trace --stats
#   gpurun_out/<tag>/pmc_*/                          separate --pmc passes (never combined with a trace domain other
#                                                    than --kernel-trace; FETCH_SIZE and WRITE_SIZE cannot share a pass:
#                                                    rocprofv3 aborts and then hangs - hence the timeouts), summarised
#                                                    by tools/pmc_summary.py
tag=${1:-prof}
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
out=gpurun_out/$tag; mkdir -p $out
if [ -z "$SKIP_BENCH" ]; then python bench.py > $out/bench.json 2> $out/bench.err; tail -c 600 $out/bench.json; echo; fi
B="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-legs --check-sample 0"
timeout -k 5 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_stats -o b -- $B > $out/prof_stats.log 2>&1
for c in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" \
         "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS" \
         "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT" \
         "SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_LDS SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL"; do
  n=$(echo $c | cut -d" " -f1)
  timeout -k 5 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/pmc_$n -o b -- $B > $out/pmc_$n.log 2>&1
done
find $out -name "*.db" -delete
python tools/pmc_summary.py $out $tag > $out/summary.json 2>&1; tail -25 $out/summary.json
