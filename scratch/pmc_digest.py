import csv, sys, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if "gemm" in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in acc.items(): print(k, "avg/launch %.4g" % (sum(v) / len(v)), "n", len(v))
