#!/bin/bash
# SQ counters of one kernel of the forward at a given batch: scratch/kpmc2.sh <tag> <kernel substring> <scenes per gpu>
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/$1; mkdir -p $O; export TMPDIR=/tmp; cd /tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/pmc$i -o p -- python $R/bench.py --scenes-per-gpu $3 --steps 4 --warmup 2 --no-cpu-baseline --no-passes > $O/pmc$i.log 2>&1
done
python $R/tools/pmc_digest.py $O $2
