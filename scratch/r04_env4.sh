#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
run() { echo "== [$1]: $(env $1 timeout 600 python bench.py --no-cpu-baseline --no-passes 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("%.0f scenes/s  %.4f ms/step" % (d["value"], d["ms_per_step"]))')"; }
for i in 1 2 3; do
for v in "X=0" "PTX_FPS_ONE=1" "PTX_SEL_ALONE=1" "PTX_TAGS_GATED=1" "PTX_FPS_ONE=1 PTX_TAGS_GATED=1"; do run "$v"; done; done
