#!/bin/bash
L=proxytransformation_amd/libproxyt_hip.so; cp $L /tmp/real.so
for spec in "cfg2 5" "cfg2 6" "cfg2 7" "cfg2 8" "cfg2 10" "cfg4 2" "cfg4 3"; do set -- $spec
for rep in 1 2; do for v in real lite256; do
  if [ $v = real ]; then cp /tmp/real.so $L; else cp scratch/lab/lib_$v.so $L; fi
  echo "== $v $1 $2 scenes: $(python bench.py --config $1 --scenes-per-gpu $2 --no-cpu-baseline --no-passes 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("%.0f scenes/s  %.4f ms/step [%s]" % (d["value"], d["ms_per_step"], " ".join("%.0f"%x for x in d["timed_blocks"]["values"])))')"
done; done; done
cp /tmp/real.so $L
