#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for k in "other_head and d256_h4_bf16" "other_head and d512_h16_f16" "other_head and d512_h16_f32" "heads4_f32" "embed512_heads16_f32"; do
  echo "=== $k"
  timeout 600 python -X faulthandler -m pytest tests/test_gpu_workloads.py tests/test_gpu_train.py -m gpu -x -q -k "$k" 2>&1 | grep -v "site-packages\|dist-packages/_pytest\|dist-packages/pluggy\|runpy" | head -40
done
