#!/bin/bash
# kernel durations of the linear kernel at a list of shapes: scratch/gemm_prof.sh "R N K gelu" ...
R=${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
cd /tmp
for sh in "$@"; do
  d=/tmp/gp_$(echo $sh | tr ' ' _)_$RANDOM
  timeout 120 rocprofv3 --kernel-trace --output-format csv -d $d -o t -- python $R/scratch/gemm_one.py $sh > /tmp/gp.log 2>&1 || { echo "$sh :: failed"; tail -3 /tmp/gp.log; continue; }
  f=$(find $d -name "*kernel_trace.csv" | head -1)
  python3 - "$f" "$sh" <<'PY'
import csv, sys
v = sorted(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in csv.DictReader(open(sys.argv[1])) if "gemm" in r["Kernel_Name"])
print(sys.argv[2], "::", "med %.2f us  min %.2f us  n %d" % (v[len(v)//2] / 1e3, v[0] / 1e3, len(v)))
PY
done
