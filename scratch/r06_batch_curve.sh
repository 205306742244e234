#!/bin/bash
# r06: scenes/s against scenes per call (one GPU): cfg2 with bf16 and with fp32 features, cfg4, cfg4_room
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
run() { echo "$1 $3 scenes/call=$2: $(timeout 600 python bench.py --no-cpu-baseline --no-passes --config $1 --scenes-per-gpu $2 ${3:+--img-dtype $3} ${4:-} 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("%.0f scenes/s  %.4f ms/step" % (d["value"], d["ms_per_step"]))')"; }
for b in 1 2 3 4 6 8 12 16; do run cfg2 $b bf16; done
for b in 24 32; do run cfg2 $b bf16 "--steps 20"; done
for b in 1 2 4 8 16; do run cfg2 $b f32; done
for b in 1 2 3 4 6 8 12 16 24; do run cfg4 $b; done
for b in 1 6 12; do run cfg4_room $b; done
for b in 1 4 8 16; do run cfg5 $b f16 "--steps 10 --warmup 3 --repeats 3 --setup-forwards 6"; done
