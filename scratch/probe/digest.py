import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
d = collections.defaultdict(list); gaps = collections.defaultdict(list)
seen = collections.Counter(); prev = None
for r in rows:
    n = r["Kernel_Name"].split("(")[0]; seen[n] += 1
    phase = "hot" if seen[n] <= 20 else "cold"
    d[(n, phase)].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    if prev is not None: gaps[(n, phase)].append(int(r["Start_Timestamp"]) - prev)
    prev = int(r["End_Timestamp"])
for k in sorted(d):
    v = sorted(d[k]); g = sorted(gaps[k])
    print(sys.argv[2], k, "dur med %d min %d" % (v[len(v)//2], v[0]), "gap-before med %d" % (g[len(g)//2] if g else -1))
