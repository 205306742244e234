#!/bin/bash
# training step: wall time, rocprofv3 kernel statistics and one step's kernel trace -> gpurun_out/r04train (copied to profiles/ by hand)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r04train; mkdir -p $O; export TMPDIR=/tmp; cd $R
python scratch/train_time.py > $O/r04_train_time.txt 2>&1
LOSS=1 ONLY_STEP=1 python scratch/train_time.py 2>&1 | grep "train step" | sed 's/train step/train step with a scalar loss built from the outputs (LOSS=1)/' >> $O/r04_train_time.txt
python scratch/train_hostprof4.py 2>&1 | grep -v amdgpu.ids >> $O/r04_train_time.txt
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/st -o k -- python $R/scratch/train_time.py > $O/stats.log 2>&1
python - <<PY
import csv, glob, re, collections
f = glob.glob("$O/st/**/k_kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
with open("$O/r04_train_kernel_stats.csv", "w", newline="") as g:
    w = csv.writer(g); w.writerow(["kernel", "calls", "total_us", "avg_us", "percent"])
    for r in rows:
        n = r["Name"]; n = n[:n.find("(")] if "(" in n else n
        w.writerow([n.replace("ptx::", "").replace("void ", ""), r["Calls"], f'{float(r["TotalDurationNs"])/1e3:.1f}', f'{float(r["AverageNs"])/1e3:.2f}', r["Percentage"]])
f = glob.glob("$O/st/**/k_kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
tf = [i for i, n in enumerate(names) if "k_select<" in n]
a, b = tf[-3], tf[-2]
t0 = int(rows[a]["Start_Timestamp"])
with open("$O/r04_train_step_trace.txt", "w") as g:
    g.write("# one training step under rocprofv3 --kernel-trace (start us, end us, duration us, queue, kernel, grid); the tracer slows the host, stream overlap is NOT representative\n")
    for r in rows[a:b]:
        n = r["Kernel_Name"]; n = n[:n.find("(")] if "(" in n else n
        n = n.replace("ptx::", "").replace("void ", "")[:60]
        s = (int(r["Start_Timestamp"]) - t0) / 1e3; e = (int(r["End_Timestamp"]) - t0) / 1e3
        g.write(f"{s:9.1f} {e:9.1f} {e-s:7.1f} q{r['Queue_Id']} {n} grid={r['Grid_Size_X']}x{r['Grid_Size_Y']}x{r['Grid_Size_Z']}\n")
PY
rm -rf $O/st
cat $O/r04_train_time.txt; head -12 $O/r04_train_kernel_stats.csv
