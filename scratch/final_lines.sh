#!/bin/bash
# extra bench lines kept under profiles/ (run on the GPU box after tools/profile_round.sh)
O=gpurun_out/r02_final; mkdir -p $O
python bench.py --scenes-per-gpu 32 --no-cpu-baseline --no-passes > $O/bench_b32.json 2>/dev/null
python bench.py --config cfg1 --no-cpu-baseline --no-passes > $O/bench_cfg1.json 2>/dev/null
python bench.py --config cfg4 --no-cpu-baseline --no-passes > $O/bench_cfg4.json 2>/dev/null
python bench.py --config cfg5 --no-cpu-baseline --no-passes > $O/bench_cfg5.json 2>/dev/null
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_args.json 2>/dev/null
for f in $O/bench_*.json; do python - $f <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[1], d["value"], d["ms_per_step"], d["roofline"]["frac"] if d.get("roofline") else None)
PY
done
