#!/bin/bash
# lanes (several forwards in flight, bench.py --streams) at the shapes whose step is the farthest point sampling
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
run() { echo "== $1: $(timeout 600 python bench.py --no-cpu-baseline --no-passes $2 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); b=d.get("blocks") or {}; print("%.0f scenes/s  %.4f ms/step  %s" % (d["value"], d["ms_per_step"], {k: b[k] for k in b if k in ("min","max","median","value_min","value_max")}))')"; }
for i in 1 2; do
run "cfg4 b1 1 lane" "--config cfg4"
run "cfg4 b1 2 lanes" "--config cfg4 --streams 2"
run "cfg4 b1 2 lanes single-thread" "--config cfg4 --streams 2 --single-thread"
run "cfg4 b1 3 lanes" "--config cfg4 --streams 3"
run "cfg4 b1 3 lanes single-thread" "--config cfg4 --streams 3 --single-thread"
run "cfg4 b6 2 lanes single-thread" "--config cfg4 --scenes-per-gpu 6 --streams 2 --single-thread"
run "cfg4 b6 3 lanes single-thread" "--config cfg4 --scenes-per-gpu 6 --streams 3 --single-thread"
run "cfg5 b1 1 lane" "--config cfg5"
run "cfg5 b1 2 lanes" "--config cfg5 --streams 2"
run "cfg5 b1 3 lanes" "--config cfg5 --streams 3"
run "cfg5 b1 4 lanes" "--config cfg5 --streams 4"
run "cfg5 b1 4 lanes single-thread" "--config cfg5 --streams 4 --single-thread"
run "cfg1 1 lane" "--config cfg1"
run "cfg1 2 lanes" "--config cfg1 --streams 2"
run "cfg1 2 lanes single-thread" "--config cfg1 --streams 2 --single-thread"
done
run "cfg5 b16 2 lanes single" "--config cfg5 --scenes-per-gpu 16 --steps 10 --warmup 3 --repeats 3 --setup-forwards 6 --streams 2 --single-thread"
run "cfg5 b8 1 lane" "--config cfg5 --scenes-per-gpu 8 --steps 10 --warmup 3 --repeats 3 --setup-forwards 6"
run "cfg5 b8 2 lanes single" "--config cfg5 --scenes-per-gpu 8 --steps 10 --warmup 3 --repeats 3 --setup-forwards 6 --streams 2 --single-thread"
