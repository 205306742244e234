#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
run() { echo "== [$1] $2: $(env $1 timeout 600 python bench.py --no-cpu-baseline --no-passes $3 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("%.0f scenes/s  %.4f ms/step" % (d["value"], d["ms_per_step"]))')"; }
for v in "X=0" "PTX_IMG_AFTER_CLUSTER=0" "PTX_EARLY_PROXIES=1"; do
  run "$v" "cfg5 b16" "--config cfg5 --scenes-per-gpu 16 --steps 10 --warmup 3 --repeats 3 --setup-forwards 6"
  run "$v" "cfg5 b4" "--config cfg5 --scenes-per-gpu 4 --steps 10 --warmup 3 --repeats 3 --setup-forwards 6"
  run "$v" "cfg5 b2" "--config cfg5 --scenes-per-gpu 2 --steps 20 --warmup 3 --repeats 3 --setup-forwards 6"
done
run "X=0" "cfg4 b6" "--config cfg4 --scenes-per-gpu 6"
run "X=0" "cfg4 b4" "--config cfg4 --scenes-per-gpu 4"
run "PTX_EARLY_PROXIES=0" "cfg4 b4" "--config cfg4 --scenes-per-gpu 4"
