#!/bin/bash
# interleaved A/B of two environment settings through bench.py:  env_ab.sh "VAR=a" "VAR=b" <scenes> [reps]
A=$1; B=$2; SC=$3; N=${4:-2}
for rep in $(seq $N); do for v in "$A" "$B"; do
  echo "== [$v] $SC scenes/GPU: $(env $v timeout 300 python bench.py --no-cpu-baseline --no-passes --scenes-per-gpu $SC 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); r=d["roofline"]; print("%.0f scenes/s  %.4f ms/step  pool %.1f us (%.3f)" % (d["value"], d["ms_per_step"], r["avg_launch_us"], r["frac"]))')"
done; done
