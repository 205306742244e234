#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r04t1; mkdir -p $O; export TMPDIR=/tmp; cd /tmp
python $R/scratch/train_gemm_shapes.py > $O/shapes.txt 2>&1
rocprofv3 --kernel-trace --output-format csv -d $O/tr -o k -- python $R/scratch/train_time.py > $O/tr.log 2>&1
python - <<PY
import csv, glob
f = glob.glob("$O/tr/**/k_kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# find the last k_tokens_finish_bwd (one per train step) and print the step before it
idx = [i for i, r in enumerate(rows) if "k_grid" in r["Kernel_Name"] or "k_minmax" in r["Kernel_Name"]]
names = [r["Kernel_Name"] for r in rows]
tf = [i for i, n in enumerate(names) if "k_tokens_finish_bwd" in n]
# a train step: from the k_minmax before tf[-2] .. the k_minmax before... take window between two consecutive tokens_finish_bwd
a, b = tf[-3], tf[-2]
t0 = int(rows[a]["Start_Timestamp"])
with open("$O/step_trace.txt", "w") as g:
    for r in rows[a:b]:
        n = r["Kernel_Name"]; n = n[:n.find("(")] if "(" in n else n
        n = n.replace("ptx::", "").replace("void ", "")[:60]
        s = (int(r["Start_Timestamp"]) - t0) / 1e3; e = (int(r["End_Timestamp"]) - t0) / 1e3
        g.write(f"{s:9.1f} {e:9.1f} {e-s:7.1f} q{r['Queue_Id']} {n} grid={r['Grid_Size_X']}x{r['Grid_Size_Y']}x{r['Grid_Size_Z']}\n")
PY
rm -rf $O/tr
tail -3 $O/shapes.txt; wc -l $O/step_trace.txt
