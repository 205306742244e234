#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r04t2; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_train.py -x -q > $O/test.txt 2>&1; tail -15 $O/test.txt
timeout 600 python scratch/train_time.py > $O/time.txt 2>&1; cat $O/time.txt
