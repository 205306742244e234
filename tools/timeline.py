"""Print the GPU timeline of one steady-state forward from a rocprofv3 kernel trace CSV
(tools/profile_round.sh -> stats_<dtype>/k_kernel_trace.csv): start/end relative to the step's
first kernel, per kernel, so the critical chain (image branch vs clustering chain) can be read off."""
import csv
import sys


def main():
    fn, step = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else -1
    rows = [r for r in csv.DictReader(open(fn))]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    # a step starts at each k_img_mean* launch
    starts = [i for i, r in enumerate(rows) if "k_img_mean" in r["Kernel_Name"]]
    # the profiler makes some steps host-bound: take the shortest step among the last ones (the GPU-bound steady state)
    cand = range(max(1, len(starts) - 40), len(starts) - 1) if step < 0 else [step]
    step = min(cand, key=lambda i: int(rows[starts[i + 1]]["Start_Timestamp"]) - int(rows[starts[i]]["Start_Timestamp"]))
    a, b = starts[step], starts[step + 1]
    t0 = int(rows[a]["Start_Timestamp"])
    prev_end = None
    for r in rows[a:b]:
        name = r["Kernel_Name"].replace("ptx::", "").replace("void ", "")
        name = name[:name.find("(")] if "(" in name else name
        s, e = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
        print(f"{s:8.1f} {e:8.1f} {e - s:7.1f}  q{r.get('Queue_Id', '?'):>2}  {name}")
    print("step period us:", (int(rows[b]["Start_Timestamp"]) - t0) / 1e3)


if __name__ == "__main__":
    main()
