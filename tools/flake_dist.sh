#!/bin/bash
# Repeats the two-ranks-on-one-GPU tests to catch intermittent failures; full logs of failing runs under gpurun_out/flake/.
# usage: tools/flake_dist.sh [runs] [pytest -k expression]
N=${1:-12}; K=${2:-two_ranks}
mkdir -p gpurun_out/flake; : > gpurun_out/flake/summary.log
for i in $(seq 1 $N); do
  L=gpurun_out/flake/run_$i.log
  timeout 600 python -m pytest tests/test_dist_gpu.py -k "$K" -q -m gpu -p no:cacheprovider -W always > $L 2>&1
  echo "$i $(grep -E 'passed|failed' $L | tail -1) $(grep -c 're-running once' $L) first attempt(s) re-run" >> gpurun_out/flake/summary.log
  if ! grep -q "failed\|re-running once" $L; then rm $L; fi
done
cat gpurun_out/flake/summary.log
