#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out/r04j; mkdir -p $O
run() { echo "== $*: $(timeout 300 python bench.py --no-cpu-baseline --no-passes $* 2>$O/err.txt | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); r=d["roofline"]; print("%.0f scenes/s  %.4f ms/step  pool %.1f us (%.3f) n=%d" % (d["value"], d["ms_per_step"], r["avg_launch_us"], r["frac"], r["launches"]))' 2>&1 | tail -1)"; }
{ run --time-every 4; run --time-every 1; run --time-every 4; run --time-every 1; run --time-every 1000000; } 2>&1 | tee $O/timing.txt
export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o k -- python $R/bench.py --steps 100 --warmup 10 --repeats 1 --no-passes --no-cpu-baseline > $O/stats.log 2>&1
grep -i "k_img_pool" $(find $O/stats -name '*kernel_stats.csv' | head -1) | cut -c1-160
tail -1 $O/stats.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); r=d["roofline"]; print("under rocprof: bench says pool %.1f us" % r["avg_launch_us"])'
