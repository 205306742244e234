#!/bin/bash
# A/B env settings over workloads: scratch/envab2.sh "VAR=a" "VAR=b"
for args in "--config cfg4" "--config cfg1" "--config cfg5" "--config cfg2"; do
 for r in 1 2; do for e in "$@"; do
  env $e python bench.py --no-cpu-baseline --no-passes --steps 100 $args 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$args', '$e', d['value'], d['ms_per_step'])
"; done; done; done
