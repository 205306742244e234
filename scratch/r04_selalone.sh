#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out/r04o; mkdir -p $O
bash scratch/env_ab.sh "PTX_SEL_ALONE=0" "PTX_SEL_ALONE=1" 4 3 2>&1 | tee $O/ab_selalone.txt
for c in cfg4 cfg1; do for v in 0 1; do echo "== $c PTX_SEL_ALONE=$v: $(PTX_SEL_ALONE=$v timeout 300 python bench.py --config $c --no-cpu-baseline --no-passes 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("%.0f scenes/s %.4f ms" % (d["value"], d["ms_per_step"]))')"; done; done 2>&1 | tee -a $O/ab_selalone.txt
echo "== cfg4 b6"; for v in 0 1 0 1; do echo "PTX_SEL_ALONE=$v: $(PTX_SEL_ALONE=$v timeout 300 python bench.py --config cfg4 --scenes-per-gpu 6 --no-cpu-baseline --no-passes 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("%.0f scenes/s %.4f ms" % (d["value"], d["ms_per_step"]))')"; done 2>&1 | tee -a $O/ab_selalone.txt
timeout 600 python -m pytest tests/test_gpu_edge_cases.py tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -2
