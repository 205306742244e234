"""Average every PMC counter per kernel from rocprofv3 counter_collection CSVs under a directory tree.
usage: python tools/pmc_digest.py DIR [name-substring ...]"""
import csv, glob, os, sys
from collections import defaultdict
def main():
    root, subs = sys.argv[1], sys.argv[2:]
    acc = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
    for fn in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(fn)):
            k = r["Kernel_Name"]
            k = k[:k.find("(")] if "(" in k else k
            k = k.replace("void ", "").replace("ptx::", "")
            if subs and not any(s in k for s in subs):
                continue
            a = acc[k][r["Counter_Name"]]
            a[0] += 1; a[1] += float(r["Counter_Value"])
    for k in sorted(acc):
        print(k)
        for c in sorted(acc[k]):
            n, v = acc[k][c]
            print(f"    {c:40s} {v / n:16.1f}  (n={n})")
if __name__ == "__main__":
    main()
