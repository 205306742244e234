#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/${1:-r04t6}; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_train.py tests/test_gpu_host.py -x -q > $O/test.txt 2>&1; tail -3 $O/test.txt
timeout 600 python scratch/train_time.py > $O/time.txt 2>&1; cat $O/time.txt
timeout 600 python scratch/train_hostprof.py > $O/hostprof.txt 2>&1; head -50 $O/hostprof.txt
