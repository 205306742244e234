#!/bin/bash
# A/B of k_img_pool builds / environment switches on one box: parity subset, then whole-step bench lines (the line's roofline object is
# the pool kernel between events) at 4 and 32 scenes per GPU.   usage: pool_ab.sh "ENV=.. ENV=.." "ENV=.." ...   ("" = shipped)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
i=0
for envs in "$@"; do
  i=$((i+1))
  echo "== variant $i: [$envs]"
  ( for kv in $envs; do export "$kv"; done
    timeout 600 python -m pytest $R/tests/test_gpu_parity.py $R/tests/test_gpu_workloads.py -m gpu -x -q -k "img or forward or workload or bench" 2>&1 | tail -2
    for sc in 4 32; do
      timeout 300 python $R/bench.py --steps 60 --warmup 12 --no-cpu-baseline --no-passes --scenes-per-gpu $sc 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']
        print('   scenes/GPU $sc: value %.0f  ms/step %.4f  pool frac %.3f achieved %.0f GB/s' % (d['value'], d['ms_per_step'], r['frac'], r['achieved']))
"
    done )
done 2>&1 | grep -v amdgpu.ids | tee $O/pool_ab.txt
