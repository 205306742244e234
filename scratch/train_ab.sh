#!/bin/bash
# interleaved A/B of the training-step time under environment settings: scratch/train_ab.sh "A=1" "B=2" ...   (3 rounds each)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for r in $(seq ${ROUNDS:-3}); do
  for cfg in "$@"; do
    echo -n "[$cfg] "; env $cfg ONLY_STEP=1 python scratch/train_time.py 2>&1 | grep "train step"
  done
done
