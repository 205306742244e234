#!/bin/bash
# on the GPU box: times of the load maps alone, then their L2 request counts (make -C scratch rowspan_bench first, here)
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
{ echo "== 784 images (4 scenes)"; $R/scratch/rowspan_bench 784; echo "== 6272 images (32 scenes)"; $R/scratch/rowspan_bench 6272
  cd /tmp && rm -rf $O/rowspan_pmc && rocprofv3 --kernel-trace --pmc TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum --output-format csv -d $O/rowspan_pmc -o p -- $R/scratch/rowspan_bench 784 > /dev/null 2>&1
  python - <<PY
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob("$O/rowspan_pmc/**/p_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[(r["Kernel_Name"][:40], r["LDS_Block_Size"] if "LDS_Block_Size" in r else "", r["Counter_Name"])].append(float(r["Counter_Value"]))
print("== L1 -> L2 read requests per launch (784 images = 1.41 M lines of 128 B)")
for k, v in sorted(acc.items()): print(f"  {k[0]:40s} lds {k[1]:>7s} {k[2]:34s} {sum(v)/len(v):14.5g}  n {len(v)}")
PY
} 2>&1 | grep -v amdgpu.ids | tee $O/rowspan.txt
