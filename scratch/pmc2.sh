export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd /tmp
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU" "SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS" "GRBM_GUI_ACTIVE TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum" "TCC_EA0_RDREQ_sum TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $R/gpurun_out/pmcy$i -o p -- python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline > $R/gpurun_out/pmcy$i.log 2>&1
  tail -2 $R/gpurun_out/pmcy$i.log | cut -c1-200
done
