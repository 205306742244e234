#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out/r04e; mkdir -p $O
for s in 1 2 3 1 2; do
  echo "== streams $s: $(timeout 300 python bench.py --no-cpu-baseline --no-passes --streams $s $* 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); r=d["roofline"]; print("%.0f scenes/s  %.4f ms/step  pool %.1f us (%.3f)  blocks %s" % (d["value"], d["ms_per_step"], r["avg_launch_us"], r["frac"], d["timed_blocks"]["values"]))')"
done 2>&1 | tee $O/streams_$(echo "$*" | tr -d ' -').txt
