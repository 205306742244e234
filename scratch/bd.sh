#!/bin/bash
# per-site breakdown for a list of libraries: scratch/bd.sh a.so b.so
for L in "$@"; do cp $L proxytransformation_amd/libproxyt_hip.so
  python bench.py --no-cpu-baseline --no-passes --steps 100 --breakdown 2>&1 | grep "^per-kernel" | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l.split(':',1)[1]); print('$L', {k.replace('k_gemm_nt','g'): round(v,1) for k,v in d.items() if 'gemm' in k or 'attn' in k or 'heads' in k})
"; done
