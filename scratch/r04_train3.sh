#!/bin/bash
# train step: tests, wall time, one-step kernel trace (gpurun_out/$1)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/${1:-r04t3}; mkdir -p $O; export TMPDIR=/tmp; cd $R
if [ -z "$SKIPTEST" ]; then timeout 900 python -m pytest tests/test_gpu_train.py -x -q > $O/test.txt 2>&1; tail -3 $O/test.txt; fi
timeout 600 python scratch/train_time.py > $O/time.txt 2>&1; cat $O/time.txt
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $O/tr -o k -- python $R/scratch/train_time.py > $O/tr.log 2>&1
python - <<PY
import csv, glob, re, collections
f = glob.glob("$O/tr/**/k_kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
tf = [i for i, n in enumerate(names) if "k_select<" in n]
a, b = tf[-3], tf[-2]
t0 = int(rows[a]["Start_Timestamp"])
tot = 0; c = collections.Counter(); cn = collections.Counter()
with open("$O/step_trace.txt", "w") as g:
    for r in rows[a:b]:
        n = r["Kernel_Name"]; n = n[:n.find("(")] if "(" in n else n
        n = n.replace("ptx::", "").replace("void ", "")[:60]
        s = (int(r["Start_Timestamp"]) - t0) / 1e3; e = (int(r["End_Timestamp"]) - t0) / 1e3
        g.write(f"{s:9.1f} {e:9.1f} {e-s:7.1f} q{r['Queue_Id']} {n} grid={r['Grid_Size_X']}x{r['Grid_Size_Y']}x{r['Grid_Size_Z']}\n")
        k = re.sub(r"at::native::.*", "torch", n); c[k] += e - s; cn[k] += 1; tot += e - s
with open("$O/step_summary.txt", "w") as g:
    g.write(f"kernels {b-a}  busy {tot:.1f} us\n")
    for k, v in c.most_common(): g.write(f"{k:44s} {cn[k]:4d} {v:8.1f}\n")
PY
rm -rf $O/tr
head -30 $O/step_summary.txt
