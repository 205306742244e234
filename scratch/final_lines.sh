#!/bin/bash
# extra bench lines kept under profiles/ (run on the GPU box after tools/profile_round.sh): scratch/final_lines.sh <tag>
TAG=${1:-r06}; O=gpurun_out/${TAG}_final; mkdir -p $O; export TMPDIR=/tmp
python bench.py --scenes-per-gpu 32 --no-cpu-baseline --no-passes > $O/bench_b32.json 2>/dev/null
python bench.py --scenes-per-gpu 1 --no-cpu-baseline --no-passes > $O/bench_b1.json 2>/dev/null
python bench.py --config cfg1 --no-cpu-baseline --no-passes > $O/bench_cfg1.json 2>/dev/null
python bench.py --config cfg4 --no-cpu-baseline --no-passes > $O/bench_cfg4.json 2>/dev/null
python bench.py --config cfg4 --scenes-per-gpu 6 --no-cpu-baseline --no-passes > $O/bench_cfg4_b6.json 2>/dev/null
python bench.py --config cfg5 --no-cpu-baseline --no-passes > $O/bench_cfg5.json 2>/dev/null
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_args.json 2>/dev/null
# the shipped configuration at the reference's training batch (6 scenes per GPU, CFG:145): kernel timeline of one step
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/stats_cfg4_b6 -o k -- \
    python $GRAFT_REPO_ROOT/bench.py --config cfg4 --scenes-per-gpu 6 --steps 60 --warmup 10 --no-cpu-baseline --no-passes > /dev/null 2>&1 )
python tools/timeline.py "$(find $O/stats_cfg4_b6 -name '*kernel_trace.csv' | head -1)" > $O/${TAG}_timeline_cfg4_b6.txt 2>/dev/null
for f in $O/bench_*.json; do python - $f <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[1], d["value"], d["ms_per_step"], d["roofline"]["frac"] if d.get("roofline") else None)
PY
done
