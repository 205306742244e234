#!/bin/bash
# r05: kernel statistics of the training step (300 steps at the training shape), per-step totals
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/trainstats
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/tr -o t -- python $R/scratch/train_long_run.py 300 1 1 1 > $O/run.log 2>&1
f=$(find $O/tr -name '*kernel_stats.csv' | head -1)
python - "$f" <<'PY' > $O/r06_train_kernel_stats.csv
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
print("kernel,calls_per_step,avg_us,us_per_step,percent")
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows:
    calls = int(r["Calls"]); t = float(r["TotalDurationNs"])
    print(f'"{r["Name"][:90]}",{calls / 305:.2f},{t / calls / 1e3:.2f},{t / 305 / 1e3:.1f},{100 * t / tot:.1f}')
print(f'"TOTAL",,,{tot / 305 / 1e3:.1f},100')
PY
python $R/scratch/apart_trace_digest.py $(find $O/tr -name '*kernel_trace.csv' | head -1) | tail -4 > $O/digest.txt
rm -rf $O/tr
head -45 $O/r06_train_kernel_stats.csv; tail -1 $O/r06_train_kernel_stats.csv; cat $O/digest.txt
