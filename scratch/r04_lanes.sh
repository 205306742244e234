#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out/r04f; mkdir -p $O
run() { echo "== $*: $(timeout 300 python bench.py --no-cpu-baseline --no-passes $* 2>$O/err.txt | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); r=d["roofline"]; print("%.0f scenes/s  %.4f ms/step  pool %.1f us (%.3f)  blocks %s" % (d["value"], d["ms_per_step"], r["avg_launch_us"], r["frac"], d["timed_blocks"]["values"]))' 2>&1 | tail -1)"; tail -2 $O/err.txt | grep -v amdgpu.ids; }
{
run --streams 1
run --streams 2
run --streams 2 --no-lane-token
run --streams 3
run --streams 4
run --streams 2 --single-thread
run --streams 1
run --streams 2
run --streams 3
} 2>&1 | tee $O/lanes.txt
