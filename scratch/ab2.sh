#!/bin/bash
# A/B two libraries over several workloads: scratch/ab2.sh a.so b.so
A=$1; B=$2
for args in "--config cfg2" "--config cfg2 --scenes-per-gpu 32" "--config cfg4" "--config cfg5" "--config cfg1"; do
 for r in 1 2; do for L in $A $B; do cp $L proxytransformation_amd/libproxyt_hip.so
  python bench.py --no-cpu-baseline --no-passes --steps 60 $args 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$args', '$L', d['value'], d['ms_per_step'])
"; done; done; done
