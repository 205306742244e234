#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out/r04m; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/gputest.txt 2>&1; tail -3 $O/gputest.txt
bash scratch/ab_interleaved.sh old real 4 3 2>&1 | tee $O/ab_reorder.txt
bash scratch/ab_interleaved.sh old real 1 1 2>&1 | tee -a $O/ab_reorder.txt
