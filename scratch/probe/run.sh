#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp; cd /tmp
for v in "$@"; do
  timeout 120 rocprofv3 --kernel-trace --output-format csv -d /tmp/pb_$v -o t -- $R/scratch/probe/probe_$v > /tmp/pb.log 2>&1 || { echo failed $v; tail -3 /tmp/pb.log; }
  f=$(find /tmp/pb_$v -name "*kernel_trace.csv" | head -1)
  python3 $R/scratch/probe/digest.py "$f" $v
done
