#!/bin/bash
# kernel durations of a python script under rocprofv3: scratch/kprof.sh <kernel-substring> script.py args...
R=${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp; cd /tmp
K=$1; shift
d=/tmp/kprof_$RANDOM
timeout 200 rocprofv3 --kernel-trace --output-format csv -d $d -o t -- python $R/"$@" > /tmp/kprof.log 2>&1 || { echo failed; tail -5 /tmp/kprof.log; exit 1; }
tail -1 /tmp/kprof.log
f=$(find $d -name "*kernel_trace.csv" | head -1)
python3 - "$f" "$K" <<'PY'
import csv, sys
v = sorted(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in csv.DictReader(open(sys.argv[1])) if sys.argv[2] in r["Kernel_Name"])
print(sys.argv[2], "med %.2f us  min %.2f us  n %d" % (v[len(v)//2] / 1e3, v[0] / 1e3, len(v)))
PY
