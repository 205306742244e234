#!/bin/bash
# rocprofv3 kernel statistics of the training step at the reference's training shape: scratch/train_prof.sh <tag>
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/$1; mkdir -p $O; export TMPDIR=/tmp; cd /tmp
python $R/scratch/train_time.py > $O/time.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o k -- python $R/scratch/train_time.py > $O/stats.log 2>&1
python - <<PY
import csv, glob
f = glob.glob("$O/stats/**/k_kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
with open("$O/train_kernel_stats.csv", "w", newline="") as g:
    w = csv.writer(g); w.writerow(["kernel", "calls", "total_us", "avg_us", "percent"])
    for r in rows:
        n = r["Name"]; n = n[:n.find("(")] if "(" in n else n
        w.writerow([n.replace("ptx::", "").replace("void ", ""), r["Calls"], f'{float(r["TotalDurationNs"])/1e3:.1f}', f'{float(r["AverageNs"])/1e3:.2f}', r["Percentage"]])
PY
cat $O/time.txt; head -40 $O/train_kernel_stats.csv
