#!/bin/bash
# interleaved A/B of two builds of the library on one box: scratch/lab/lib_old.so vs scratch/lab/lib_new.so (built in the container)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
L=proxytransformation_amd/libproxyt_hip.so
run() { cp scratch/lab/lib_$1.so $L; echo "== [$1] $2: $(timeout 600 python bench.py --no-cpu-baseline --no-passes $3 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("%.0f scenes/s  %.4f ms/step" % (d["value"], d["ms_per_step"]))')"; }
cp scratch/lab/lib_new.so $L
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_edge_cases.py tests/test_gpu_workloads.py tests/test_gpu_host.py -m gpu -q 2>&1 | tail -15
for i in 1 2; do
for v in old new; do
  run $v "cfg4 b6" "--config cfg4 --scenes-per-gpu 6"
  run $v "cfg4 b1" "--config cfg4"
  run $v "cfg5 b1" "--config cfg5"
  run $v "cfg1" "--config cfg1"
  run $v "b4" ""
  run $v "b32" "--scenes-per-gpu 32"
done; done
cp scratch/lab/lib_new.so $L
