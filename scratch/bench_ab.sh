#!/bin/bash
# A/B of library builds through bench.py on one box: bash scratch/bench_ab.sh prev real [...]   (see scratch/lab_ab.sh)
L=proxytransformation_amd/libproxyt_hip.so
cp $L /tmp/real.so
for rep in 1 2; do
for v in "$@"; do
  if [ $v = real ]; then cp /tmp/real.so $L; else cp scratch/lab/lib_$v.so $L; fi
  echo "== $v $(timeout 300 python bench.py --no-cpu-baseline --no-passes $BENCH_ARGS 2>&1 | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"])')"
done; done
cp /tmp/real.so $L
