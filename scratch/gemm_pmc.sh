#!/bin/bash
# PMC counters for one GEMM shape: scratch/gemm_pmc.sh "R N K g" CTR1 CTR2 ...   (one pass per counter)
R=${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp; cd /tmp
sh=$1; shift
for c in "$@"; do
  d=/tmp/pmc_$c
  timeout 120 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $d -o p -- python $R/scratch/gemm_one.py $sh > /tmp/pmc.log 2>&1 || { echo "$c failed"; tail -2 /tmp/pmc.log; continue; }
  f=$(find $d -name "*counter_collection.csv" | head -1)
  python3 $R/scratch/pmc_digest.py "$f"
done
