#!/bin/bash
# early proxies (point proxies + qkv of ALL clusters beside the farthest point sampling) vs the kept-rows-only path, interleaved
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
run() { echo "== [$1] $2: $(env $1 timeout 300 python bench.py --no-cpu-baseline --no-passes $3 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("%.0f scenes/s  %.4f ms/step" % (d["value"], d["ms_per_step"]))')"; }
for rep in 1 2; do for v in "PTX_EARLY_PROXIES=0" "PTX_EARLY_PROXIES=1"; do
  run "$v" "cfg4 b6" "--config cfg4 --scenes-per-gpu 6"
  run "$v" "cfg4 b1" "--config cfg4 --scenes-per-gpu 1"
  run "$v" "cfg1 b1" "--config cfg1"
  run "$v" "cfg5 b1" "--config cfg5 --steps 20 --warmup 4"
done; done
