"""Digest of a kernel trace of the training step: per step (delimited by the first kernel of the forward, k_grid... / the image pool),
the period, the GPU busy time (union of kernel intervals), the time two or more kernels overlap, per queue/stream kernel counts."""
import csv, sys
rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", ""), r.get("Stream_Id", "")))
rows.sort()
names = [r[2] for r in rows]
# a step starts at each k_grid_centers-like kernel: find the marker = first kernel name containing 'grid'
marker = next((n for n in names if "k_keep_rows" in n), None)
print("marker kernel:", marker, " kernels:", len(rows), " queues:", sorted(set(r[3] for r in rows)), " streams:", sorted(set(r[4] for r in rows)))
starts = [i for i, n in enumerate(names) if n == marker]
out = []
for a, b in zip(starts[:-1], starts[1:]):
    seg = rows[a:b]
    t0, t1 = seg[0][0], rows[b][0]
    ev = []
    for s, e, *_ in seg:
        ev.append((s, 1)); ev.append((e, -1))
    ev.sort()
    busy = ovl = 0
    depth, last = 0, ev[0][0]
    for tt, d in ev:
        if depth >= 1: busy += tt - last
        if depth >= 2: ovl += tt - last
        depth += d; last = tt
    out.append(((t1 - t0) / 1e3, busy / 1e3, ovl / 1e3, len(seg)))
print("step   period_us  busy_us  overlap_us  kernels")
for i, o in enumerate(out):
    if i % 10 == 0 or i > len(out) - 5:
        print(f"{i:4d}  {o[0]:9.1f} {o[1]:8.1f} {o[2]:9.1f} {o[3]:6d}")
import statistics as st
for lo, hi in ((0, 100), (100, 200), (200, 300)):
    s = out[lo:hi]
    if s:
        print(f"steps {lo}-{hi}: period median {st.median(x[0] for x in s):.1f} us  busy {st.median(x[1] for x in s):.1f}  overlap {st.median(x[2] for x in s):.1f}")
