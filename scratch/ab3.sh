#!/bin/bash
# A/B libraries on given configs: scratch/ab3.sh "cfg4 cfg5" a.so b.so ...
CFGS=$1; shift
for c in $CFGS; do for r in 1 2 3; do for L in "$@"; do cp $L proxytransformation_amd/libproxyt_hip.so
  python bench.py --no-cpu-baseline --no-passes --steps 100 --config $c 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$c', '$L', d['value'], d['ms_per_step'])
"; done; done; done
