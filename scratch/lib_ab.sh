#!/bin/bash
# interleaved A/B of two builds of the library on one box: scratch/lab/lib_old.so vs scratch/lab/lib_new.so (built in the container)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
L=proxytransformation_amd/libproxyt_hip.so
run() { cp scratch/lab/lib_$1.so $L; echo "== [$1] $2: $(timeout 600 python bench.py --no-cpu-baseline --no-passes $3 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("%.0f scenes/s  %.4f ms/step" % (d["value"], d["ms_per_step"]))')"; }
for i in 1 2; do
for v in old new; do
  run $v "cfg5 b16" "--config cfg5 --scenes-per-gpu 16 --steps 10 --warmup 3 --repeats 3 --setup-forwards 6"
  run $v "cfg5 b12" "--config cfg5 --scenes-per-gpu 12 --steps 10 --warmup 3 --repeats 3 --setup-forwards 6"
  run $v "cfg5 b24" "--config cfg5 --scenes-per-gpu 24 --steps 6 --warmup 2 --repeats 3 --setup-forwards 4"
done; done
cp scratch/lab/lib_new.so $L
