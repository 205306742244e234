"""Timeline of the shortest training step of a rocprofv3 kernel trace (marker: the forward's first kernel on the caller's stream,
k_minmax / grid centres): start / end / duration / queue per kernel.  usage: train_timeline.py <kernel_trace.csv>"""
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
name = lambda r: r["Kernel_Name"].replace("ptx::", "").replace("void ", "").split("(")[0]
starts = [i for i, r in enumerate(rows) if "k_ti_mean" in r["Kernel_Name"]]
cand = range(max(1, len(starts) - 30), len(starts) - 1)
k = min(cand, key=lambda i: int(rows[starts[i + 1]]["Start_Timestamp"]) - int(rows[starts[i]]["Start_Timestamp"]))
a, b = starts[k], starts[k + 1]
t0 = int(rows[a]["Start_Timestamp"])
for r in rows[a:b]:
    s, e = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
    print(f"{s:8.1f} {e:8.1f} {e - s:7.1f}  q{r.get('Queue_Id', '?'):>2}  {name(r)}")
print("step period us:", (int(rows[b]["Start_Timestamp"]) - t0) / 1e3, " kernels:", b - a)
