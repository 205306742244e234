#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_workloads.py tests/test_gpu_host.py -m gpu -x -q 2>&1 | tail -2
run() { echo "== $1: $(timeout 600 python bench.py --no-cpu-baseline --no-passes $2 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("%.0f scenes/s  %.4f ms/step" % (d["value"], d["ms_per_step"]))')"; }
run "cfg4 b8" "--config cfg4 --scenes-per-gpu 8"
run "cfg4 b6" "--config cfg4 --scenes-per-gpu 6"
run "cfg5 b8" "--config cfg5 --scenes-per-gpu 8 --steps 10 --warmup 3 --repeats 3 --setup-forwards 6"
run "cfg5 b16" "--config cfg5 --scenes-per-gpu 16 --steps 10 --warmup 3 --repeats 3 --setup-forwards 6"
run "cfg5 b1" "--config cfg5"
