#!/bin/bash
# one kernel-trace timeline of the default bench step (on the GPU box): bash scratch/tl.sh [bench args] -> gpurun_out/tl.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd /tmp
rm -rf /tmp/tl_prof
rocprofv3 --kernel-trace --output-format csv -d /tmp/tl_prof -o k -- python $R/bench.py --steps 40 --warmup 10 --no-passes --no-cpu-baseline "$@" > /tmp/tl.log 2>&1
python $R/tools/timeline.py "$(find /tmp/tl_prof -name '*kernel_trace.csv' | head -1)" > $R/gpurun_out/tl.txt 2>&1
