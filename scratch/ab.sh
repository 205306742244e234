#!/bin/bash
# A/B two builds of the library on the same box: scratch/ab.sh a.so b.so [bench args]
A=$1; B=$2; shift 2
for r in 1 2 3; do for L in $A $B; do cp $L proxytransformation_amd/libproxyt_hip.so
  python bench.py --no-cpu-baseline --no-passes --steps 200 "$@" 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$L', d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'])
"; done; done
