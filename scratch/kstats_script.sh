#!/bin/bash
# per-kernel totals of a python script under rocprofv3: scratch/kstats_script.sh script.py args...
R=${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp; cd /tmp
d=/tmp/ks_$RANDOM
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $d -o t -- python $R/"$@" > /tmp/ks.log 2>&1 || { echo failed; tail -5 /tmp/ks.log; exit 1; }
grep -v "rocprofv3\|^W2026\|^E2026" /tmp/ks.log | tail -3
f=$(find $d -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys, collections
tot = collections.defaultdict(lambda: [0, 0])
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"].replace("ptx::", "").replace("void ", "")
    n = n[:n.find("(")] if "(" in n else n
    tot[n][0] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"]); tot[n][1] += 1
s = sum(v[0] for v in tot.values())
for k, v in sorted(tot.items(), key=lambda kv: -kv[1][0])[:22]:
    print("%-60s %9.1f us  %5d calls  %5.1f %%  avg %7.1f" % (k[:60], v[0] / 1e3, v[1], 100.0 * v[0] / s, v[0] / 1e3 / v[1]))
print("total kernel time %.1f us" % (s / 1e3))
PY
