#!/bin/bash
# A/B of library builds through bench.py on one box, with the time of k_img_pool between events:  pool_var_ab.sh real nostore ...
L=proxytransformation_amd/libproxyt_hip.so
cp $L /tmp/real.so
for v in "$@"; do
  if [ $v = real ]; then cp /tmp/real.so $L; else cp scratch/lab/lib_$v.so $L; fi
  for sc in 4 32; do
    echo "== $v, $sc scenes/GPU: $(timeout 300 python bench.py --no-cpu-baseline --no-passes --scenes-per-gpu $sc 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); r=d["roofline"]; print("%.0f scenes/s  %.4f ms/step  pool %.1f us (%.3f of the HBM peak)" % (d["value"], d["ms_per_step"], r["avg_launch_us"], r["frac"]))')"
  done
done
cp /tmp/real.so $L
