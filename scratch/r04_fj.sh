#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
run() { echo "== [$1] $2: $(env $1 timeout 600 python bench.py --no-cpu-baseline --no-passes $3 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("%.0f scenes/s  %.4f ms/step" % (d["value"], d["ms_per_step"]))')"; }
for i in 1 2 3; do
for v in "PTX_FORK_EXT=0 PTX_JOIN_CHAIN=0" "PTX_FORK_EXT=1 PTX_JOIN_CHAIN=0" "PTX_FORK_EXT=0 PTX_JOIN_CHAIN=1" "PTX_FORK_EXT=1 PTX_JOIN_CHAIN=1"; do
  run "$v" "cfg4 b6" "--config cfg4 --scenes-per-gpu 6"
  run "$v" "cfg4 b3" "--config cfg4 --scenes-per-gpu 3"
done; done
