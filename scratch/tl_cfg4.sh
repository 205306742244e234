#!/bin/bash
# kernel timeline of the shortest traced step: scratch/tl_cfg4.sh "<bench args>" <tag>
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; export TMPDIR=/tmp
O=$R/gpurun_out/tl_$2; rm -rf $O; mkdir -p $O
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/st -o k -- python $R/bench.py $1 --steps 60 --warmup 10 --no-cpu-baseline --no-passes > /dev/null 2>&1 )
python tools/timeline.py "$(find $O/st -name '*kernel_trace.csv' | head -1)" > $O/timeline.txt 2>/dev/null
cat $O/timeline.txt
