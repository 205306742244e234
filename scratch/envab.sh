#!/bin/bash
# A/B environment settings on the same box: scratch/envab.sh "VAR=1" ["VAR2=.."]; baseline first
for r in 1 2; do for e in "X_=0" "$@"; do
  env $e python bench.py --no-cpu-baseline --no-passes --steps 200 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$e', d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'])
"; done; done
