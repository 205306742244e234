#!/bin/bash
# SQ counters (incl. matrix-pipe and LDS-conflict counters) of kernels of the 4-scene forward, one counter set per pass:
#   scratch/kpmc3.sh <tag> <kernel substring> [more substrings]     -> gpurun_out/<tag>/digest.txt
R=${GRAFT_REPO_ROOT:-/root/repo}; T=$1; shift; O=$R/gpurun_out/$T; mkdir -p $O; export TMPDIR=/tmp; cd /tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVES SQ_INSTS_MFMA" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/pmc$i -o p -- python $R/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-passes > $O/pmc$i.log 2>&1
done
python $R/tools/pmc_digest.py $O "$@" > $O/digest.txt
