#!/bin/bash
# scenes/s against scenes per call (one GPU) for the benchmark shape (cfg2) and the shipped configuration (cfg4): scratch/batch_curve.sh
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
run() { echo "$1 scenes/call=$2: $(timeout 600 python bench.py --no-cpu-baseline --no-passes --config $1 --scenes-per-gpu $2 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("%.0f scenes/s  %.4f ms/step" % (d["value"], d["ms_per_step"]))')"; }
for b in 1 2 3 4 6 8 12 16 24 32; do run cfg2 $b; done
for b in 1 2 3 4 6 8 12 16 24; do run cfg4 $b; done
