#!/bin/bash
# interleaved A/B of two environment settings through bench.py:  env_ab2.sh "<envA>" "<envB>" <config> <scenes> [reps] [img-dtype]
A=$1; B=$2; CFG=$3; SC=$4; N=${5:-3}; DT=${6:-}
for rep in $(seq $N); do for v in "$A" "$B"; do
  echo "== [$v] $CFG, $SC scenes/GPU: $(env $v timeout 300 python bench.py --config $CFG ${DT:+--img-dtype $DT} --no-cpu-baseline --no-passes --scenes-per-gpu $SC 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); r=d["roofline"]; print("%.0f scenes/s  %.4f ms/step  [%s]  %s %.1f us" % (d["value"], d["ms_per_step"], " ".join("%.0f"%x for x in d["timed_blocks"]["values"]), r["kernel"], r["avg_launch_us"]))')"
done; done
