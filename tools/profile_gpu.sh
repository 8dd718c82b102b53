#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel-trace stats + PMC passes for bench.py.
# Usage: tools/profile_gpu.sh <tag> [extra bench args...]
# Raw output -> gpurun_out/prof_<tag>/ ; summarise with tools/summarize_prof.py and commit
# the summary under profiles/.
TAG=${1:-r2}; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
BENCH="python $R/bench.py --steps 20 --warmup 5 --headline-only $*"
PBENCH="python $R/bench.py --steps 3 --warmup 4 --headline-only $*"   # (a drop-in graph is prepared at its second call: steady from call 3)
rocprofv3 --kernel-trace --stats -T -d $OUT/trace -o bench -f csv -- $BENCH > $OUT/trace_stdout.log 2>&1
# PMC=0 skips the counter passes (kernel trace only)
if [ "${PMC:-1}" != "0" ]; then
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" \
           "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TA_BUSY_sum" \
           "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  name=$(echo $set | tr ' ' '+' | cut -c1-40)
  rocprofv3 --pmc $set --kernel-include-regex "stream_kernel|slice_count_kernel|prologue_kernel|scale_rows_kernel|sweep_kernel|relu_fixup" -T -d $OUT/pmc_$name -o pmc -f csv -- $PBENCH > $OUT/pmc_$name.log 2>&1
done
# FETCH_SIZE / WRITE_SIZE calibration on a known 1 GiB device copy
rocprofv3 --pmc FETCH_SIZE -T -d $OUT/calib_fetch -o pmc -f csv -- python $R/tools/calib_copy.py > $OUT/calib_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -T -d $OUT/calib_write -o pmc -f csv -- python $R/tools/calib_copy.py > $OUT/calib_write.log 2>&1
fi
cd $R
python tools/summarize_prof.py $OUT > $OUT/SUMMARY.md 2> $OUT/summarize.err
cat $OUT/SUMMARY.md
