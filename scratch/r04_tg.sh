#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
run() { echo "== [$1] $2: $(env $1 timeout 600 python bench.py --no-cpu-baseline --no-passes $3 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("%.0f scenes/s  %.4f ms/step" % (d["value"], d["ms_per_step"]))')"; }
for i in 1 2 3 4 5 6; do
for v in "X=0" "PTX_TAGS_GATED=1" "PTX_FPS_ONE=1"; do run "$v" "b4" ""; done; done
for i in 1 2; do for v in "X=0" "PTX_TAGS_GATED=1"; do run "$v" "b2" "--scenes-per-gpu 2"; run "$v" "b6" "--scenes-per-gpu 6"; run "$v" "b8" "--scenes-per-gpu 8"; done; done
