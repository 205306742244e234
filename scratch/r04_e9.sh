#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
run() { echo "== [$1] $2: $(env $1 timeout 600 python bench.py --no-cpu-baseline --no-passes $3 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("%.0f scenes/s  %.4f ms/step" % (d["value"], d["ms_per_step"]))')"; }
for i in 1 2 3; do
for v in "X=0" "PTX_EARLY_PROXIES=1"; do
  run "$v" "cfg4 b1" "--config cfg4"
  run "$v" "cfg4 b2" "--config cfg4 --scenes-per-gpu 2"
  run "$v" "cfg1" "--config cfg1"
done; done
