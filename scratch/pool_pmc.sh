#!/bin/bash
# L1->L2 request counts / latency / TA occupancy of the two streaming passes over img_feat (k_img_mean16, k_img_pool):
# is the pool pass's 16 x (4 rows x 256 B at 2-B alignment) load map request-bound at the CU?
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd /tmp
i=0
for set in "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum" "TA_BUSY_avr TCP_TOTAL_ACCESSES_sum TCP_TOTAL_CACHE_ACCESSES_sum GRBM_GUI_ACTIVE" "TCP_TA_TCP_STATE_READ_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum" "TA_DATA_STALLED_BY_TC_CYCLES_sum TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum TA_BUFFER_READ_WAVEFRONTS_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/ppmc$i -o p -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-passes > $O/ppmc$i.log 2>&1
done
python - <<PY
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob("$O/ppmc*/**/p_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        k = "pool" if "k_img_pool" in n else "mean16" if "k_img_mean16" in n else None
        if k: acc[(k, r["Counter_Name"])].append(float(r["Counter_Value"]))
for (k, c), v in sorted(acc.items()):
    print(f"{k:7s} {c:40s} avg/launch {sum(v)/len(v):14.5g}  n {len(v)}")
PY
