#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out/r04l; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/gputest.txt 2>&1; tail -3 $O/gputest.txt
bash scratch/env_ab.sh "PTX_TAGS_GATED=0" "PTX_TAGS_GATED=1" 4 3 2>&1 | tee $O/ab_tags.txt
bash scratch/env_ab.sh "PTX_TAGS_GATED=0" "PTX_TAGS_GATED=1" 8 1 2>&1 | tee -a $O/ab_tags.txt
bash scratch/env_ab.sh "PTX_TAGS_GATED=0" "PTX_TAGS_GATED=1" 32 1 2>&1 | tee -a $O/ab_tags.txt
export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o k -- python $R/bench.py --steps 100 --warmup 10 --repeats 1 --no-passes --no-cpu-baseline > $O/stats.log 2>&1
python $R/tools/timeline.py "$(find $O/stats -name '*kernel_trace.csv' | head -1)" | tee $O/timeline_bf16.txt
