#!/bin/bash
# roofline_passes (attention / GEMM MFMA fractions) for a list of libraries
for L in "$@"; do cp $L proxytransformation_amd/libproxyt_hip.so
  python bench.py --no-cpu-baseline --steps 30 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l)
        for p in d['roofline_passes']:
            print('$L', 'B', p['scenes_per_gpu'], 'attn', p['proxy_attention_mfma']['us'], p['proxy_attention_mfma']['frac_of_f32_mfma_peak'], 'step', d['ms_per_step'])
"; done
