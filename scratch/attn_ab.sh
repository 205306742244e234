#!/bin/bash
# fused vs two-launch proxy attention in the whole step: scratch/attn_ab.sh
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for cfgB in "cfg2 4" "cfg2 8" "cfg2 16" "cfg2 32" "cfg4 1" "cfg4 6" "cfg1 1"; do set -- $cfgB
  for f in 0 1; do
    PTX_ATTN_FUSED=$f timeout 200 python bench.py --config $1 --scenes-per-gpu $2 --steps 40 --warmup 8 --no-cpu-baseline --no-passes 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1 B=$2 fused=$f', d['value'], d['ms_per_step'])"
  done
done
