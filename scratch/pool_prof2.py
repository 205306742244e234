import sys, numpy as np
d = np.loadtxt(sys.argv[1], dtype=np.float64)
u, ts = d[:, 0], d[:, 1:]
t0 = ts[:, 0].min()
print("tile0 %.0f  reload-issue %.0f  tile1 %.0f  total %.0f cycles (mean per wave)" % ((ts[:,1]-ts[:,0]).mean(), (ts[:,2]-ts[:,1]).mean(), (ts[:,3]-ts[:,2]).mean(), (ts[:,3]-ts[:,0]).mean()))
st = np.array([ts[u == k, 0].min() for k in np.unique(u)]) - t0
en = np.array([ts[u == k, 3].max() for k in np.unique(u)]) - t0
print("span %.0f cycles = %.1f us @2.35GHz; start pct 10/50/90: %s; end pct 50/90/100: %s" % (en.max(), en.max()/2350, np.percentile(st,[10,50,90]), np.percentile(en,[50,90,100])))
print("concurrency (units alive at mid-span):", ((st < en.max()/2) & (en > en.max()/2)).sum())
for frac in (0.1, 0.25, 0.5, 0.75, 0.9):
    t = en.max() * frac
    print("  alive at %.0f%%: %d" % (frac*100, ((st <= t) & (en > t)).sum()))
