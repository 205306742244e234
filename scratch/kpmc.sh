#!/bin/bash
# PMC counters for one kernel of a bench run: scratch/kpmc.sh <kernel-substring> "<bench args>" CTR...
R=${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp; cd /tmp
K=$1; A=$2; shift 2
for c in "$@"; do
  d=/tmp/kp_$c
  timeout 200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $d -o p -- python $R/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-passes $A > /tmp/kp.log 2>&1 || { echo "$c failed"; tail -2 /tmp/kp.log; continue; }
  f=$(find $d -name "*counter_collection.csv" | head -1)
  python3 - "$f" "$K" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if sys.argv[2] in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in acc.items(): print(k, "avg/launch %.4g" % (sum(v) / len(v)), "n", len(v))
PY
done
