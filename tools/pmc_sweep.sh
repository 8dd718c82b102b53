#!/bin/bash
# Runs on the GPU box: PMC passes (L2 hit rate, fabric bytes) of tools/sweep.py for one configuration.
# Usage: tools/pmc_sweep.sh <tag> <sweep.py args...>     -> gpurun_out/pmc_<tag>/SUMMARY.md
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_EA0_ATOMIC_sum"; do
  name=$(echo $set | tr ' ' '+' | cut -c1-40)
  rocprofv3 --pmc $set --kernel-include-regex "agg_kernel|stream_kernel|slice_count|prologue" -T -d $OUT/$name -o pmc -f csv -- python $R/tools/sweep.py --steps 2 "$@" > $OUT/$name.log 2>&1
done
cd $R
python tools/summarize_prof.py $OUT > $OUT/SUMMARY.md 2> $OUT/summarize.err
cat $OUT/SUMMARY.md
