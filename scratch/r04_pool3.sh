#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out/r04k; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_workloads.py tests/test_gpu_edge_cases.py tests/test_gpu_host.py -m gpu -x -q -k "not bench_multi" > $O/gputest.txt 2>&1; tail -3 $O/gputest.txt
bash scratch/ab_interleaved.sh old real 4 3 2>&1 | tee $O/ab_pool3.txt
bash scratch/ab_interleaved.sh old real 32 1 2>&1 | tee -a $O/ab_pool3.txt
