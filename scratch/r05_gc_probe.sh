#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for i in 1 2; do
  echo "== default"; timeout 200 python scratch/train_long_run.py 500 1 1 1 2>&1 | grep "steps" | cut -c1-150
  echo "== gc.freeze() after warm-up"; GC_FREEZE=1 timeout 200 python scratch/train_long_run.py 500 1 1 1 2>&1 | grep "steps" | cut -c1-150
done
