import sys, numpy as np
d = np.loadtxt(sys.argv[1], dtype=np.int64)
u, w, ts = d[:, 0], d[:, 1], d[:, 2:]
t0 = ts[:, 0].min()
names = ["start->loads issued+prologue", "barrier1 wait", "stage1 (wait loads + scores)", "barrier2 wait", "stage2 softmax", "barrier3 wait", "stage3 gather"]
dt = np.diff(ts, axis=1)
print("cycles per phase (mean / median / p90) over %d waves" % len(d))
for k, nm in enumerate(names):
    print(f"  {nm:32s} {dt[:, k].mean():9.0f} {np.median(dt[:, k]):9.0f} {np.percentile(dt[:, k], 90):9.0f}")
tot = ts[:, 7] - ts[:, 0]
print("  total per wave                   %9.0f %9.0f" % (tot.mean(), np.median(tot)))
span = ts[:, 7].max() - t0
print("kernel span (cycles):", span, " units:", len(np.unique(u)))
# start time distribution of units
st = np.array([ts[u == k, 0].min() for k in np.unique(u)]) - t0
en = np.array([ts[u == k, 7].max() for k in np.unique(u)]) - t0
print("unit lifetime mean:", (en - st).mean(), " start times pct 25/50/75:", np.percentile(st, [25, 50, 75]))
