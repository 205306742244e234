#!/bin/bash
# interleaved A/B of two library builds through bench.py (box drift cancels):  ab_interleaved.sh <tagA> <tagB> <scenes> [reps]
L=proxytransformation_amd/libproxyt_hip.so; cp $L /tmp/real.so
A=$1; B=$2; SC=$3; N=${4:-3}
for rep in $(seq $N); do for v in $A $B; do
  if [ $v = real ]; then cp /tmp/real.so $L; else cp scratch/lab/lib_$v.so $L; fi
  echo "== $v, $SC scenes/GPU: $(timeout 300 python bench.py --no-cpu-baseline --no-passes --scenes-per-gpu $SC 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); r=d["roofline"]; print("%.0f scenes/s  %.4f ms/step  pool %.1f us (%.3f)" % (d["value"], d["ms_per_step"], r["avg_launch_us"], r["frac"]))')"
done; done
cp /tmp/real.so $L
