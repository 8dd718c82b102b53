#!/bin/bash
# The whole -m gpu suite under process-wide GNNA_TUNE settings (tests that compare the library's own choices skip when
# a schedule is forced), then the seeded fuzz at a higher case count.  Output: gpurun_out/<tag>/knob_soak.log
TAG=${1:-soak}; R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out/$TAG; L=gpurun_out/$TAG/knob_soak.log; : > $L
for T in "DET=1" "SWEEP=1,PHASES=8" "PHASES=32" "SWEEP=1,PHASES=24,SLACK=1,BPC=2" "ZERO=1,G=1" "G=5,PHASES=16" "BLOCKS=1" "BLOCKS=1,PHASES=3,PACK=1" "PRESCALE=2,PAD=2,BLOCKS=2" "PACK=1,U=8" "CHECK=1,PACK=1" "CHECK=3,PHASES=12"; do
  echo "== GNNA_TUNE=$T" >> $L
  GNNA_TUNE=$T timeout 1200 python -m pytest tests -q -m gpu -p no:cacheprovider --deselect tests/test_regret_gpu.py --deselect tests/test_rccl_gpu.py 2>&1 | grep -E "passed|failed|FAILED|error" | head -8 >> $L
done
for S in 11 12 13; do
  echo "== fuzz GNNA_TEST_CASES=200 GNNA_TEST_SEED=$S" >> $L
  GNNA_TEST_CASES=200 GNNA_TEST_SEED=$S timeout 1200 python -m pytest tests/test_fuzz_gpu.py -q -m gpu -p no:cacheprovider 2>&1 | grep -E "passed|failed|FAILED|error" | head -5 >> $L
done
cat $L
