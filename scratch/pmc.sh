export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd /tmp
i=0
for set in "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum" "TA_BUSY_avr TCP_TOTAL_ACCESSES_sum TCP_TCR_TCP_STALL_CYCLES_sum GRBM_GUI_ACTIVE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_TAG_STALL_sum TCC_REQ_sum"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $R/gpurun_out/pmcx$i -o p -- python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline > $R/gpurun_out/pmcx$i.log 2>&1
done
ls $R/gpurun_out/pmcx*/
