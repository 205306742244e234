#!/bin/bash
# k_mlp LITE (two work-groups per CU) vs the one-work-group form, interleaved, at the shapes with more work-groups than CUs
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
run() { echo "== [$1] $2: $(env $1 timeout 300 python bench.py --no-cpu-baseline --no-passes $3 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("%.0f scenes/s  %.4f ms/step" % (d["value"], d["ms_per_step"]))')"; }
for rep in 1 2; do
  for v in "PTX_MLP_LITE=0" "PTX_MLP_LITE=1"; do
    run "$v" "cfg4 b6" "--config cfg4 --scenes-per-gpu 6"
    run "$v" "cfg2 b16" "--scenes-per-gpu 16"
    run "$v PTX_MLP_RMAX=9000" "cfg2 b32 (fused forced)" "--scenes-per-gpu 32"
    run "$v" "cfg2 b8" "--scenes-per-gpu 8"
  done
done
run "PTX_MLP_RMAX=6144" "cfg2 b32 (two GEMM launches)" "--scenes-per-gpu 32"
