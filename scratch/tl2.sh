#!/bin/bash
# timeline of a script that launches k_img_mean* once per iteration (on the GPU box): bash scratch/tl2.sh scratch/imgchain_tl.py
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd /tmp
rm -rf /tmp/tl_prof
rocprofv3 --kernel-trace --output-format csv -d /tmp/tl_prof -o k -- python $R/$1 > /tmp/tl.log 2>&1
tail -3 /tmp/tl.log > $R/gpurun_out/tl.txt
python $R/tools/timeline.py "$(find /tmp/tl_prof -name '*kernel_trace.csv' | head -1)" >> $R/gpurun_out/tl.txt 2>&1
