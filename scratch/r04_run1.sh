#!/bin/bash
# r04 GPU run 1: full GPU test suite, A/B of the gate fold and the fused bounding boxes, a kernel-trace timeline
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
O=$R/gpurun_out/r04a; mkdir -p $O
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -x -q > $O/gputest.txt 2>&1; echo "pytest rc=$?" >> $O/gputest.txt
tail -5 $O/gputest.txt
bash scratch/env_ab.sh "PTX_GATE_FOLD=0 PTX_MM_FUSE=0" "PTX_GATE_FOLD=1 PTX_MM_FUSE=0" 4 2 > $O/ab_fold.txt 2>&1
bash scratch/env_ab.sh "PTX_MM_FUSE=0" "PTX_MM_FUSE=1" 4 2 > $O/ab_mmfuse.txt 2>&1
cat $O/ab_fold.txt $O/ab_mmfuse.txt
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o k -- python $R/bench.py --steps 100 --warmup 10 --repeats 1 --no-passes --no-cpu-baseline > $O/stats.log 2>&1
python $R/tools/timeline.py "$(find $O/stats -name '*kernel_trace.csv' | head -1)" > $O/timeline_bf16.txt 2>&1
cat $O/timeline_bf16.txt
cd $R; python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err; tail -c 1500 $O/bench.json
