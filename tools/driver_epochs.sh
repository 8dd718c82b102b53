#!/bin/bash
# driver epochs (forward + backward + Adam) through main.py, the reference's loop: 10 dry runs + timed epochs
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
run() { echo "== $*"; python -m gnnadvisor_osdi21_amd.main --manual_mode False --num_epoches 50 "$@" 2>/dev/null | grep "Time (ms)"; }
run --synthetic reddit-like --dim 602 --hidden 64 --classes 41 --model gcn
run --synthetic reddit-like --dim 602 --hidden 64 --classes 41 --model gin
run --synthetic products-like --dim 100 --hidden 64 --classes 47 --model gcn
run --synthetic products-like --dim 100 --hidden 64 --classes 47 --model gin
run --synthetic amazon0505-like --dim 96 --hidden 16 --classes 22 --model gcn
run --synthetic amazon0505-like --dim 96 --hidden 64 --classes 22 --model gin
run --synthetic reddit-like --dim 602 --hidden 16 --classes 41 --model gcn
run --synthetic reddit-like --dim 602 --hidden 64 --classes 41 --model gcn --hip_graph True
run --synthetic cora-like --dim 1433 --hidden 16 --classes 7 --model gcn
