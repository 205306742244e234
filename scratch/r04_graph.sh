#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out/r04h; mkdir -p $O
run() { echo "== $*: $(timeout 400 python bench.py --no-cpu-baseline --steps 40 --repeats 3 $* 2>$O/err.txt | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("eager %.0f scenes/s (%.4f ms)   graph %s (%s ms)   %s" % (d["value"], d["ms_per_step"], d.get("value_graph_replay"), d.get("graph_replay", {}).get("ms_per_step"), d.get("graph_replay", {}).get("error", "")))' 2>&1 | tail -1)"; }
{
run --config cfg2
run --config cfg2 --scenes-per-gpu 1
run --config cfg1
run --config cfg4
run --config cfg4 --scenes-per-gpu 6
} 2>&1 | tee $O/graph.txt
