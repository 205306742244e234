#!/bin/bash
# PMC passes over k_gemm128x / k_gemm64x alone (run on the GPU box): scratch/gemm128_pmc.sh <tag> "<R N K>" [policy]
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/$1; mkdir -p $O; export TMPDIR=/tmp; cd /tmp
SH=${2:-"16384 2048 512"}; POL=${3:-1}
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVES" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_ACTIVE_INST_MISC SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY" \
           "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCP_TCC_READ_REQ_sum TCC_REQ_sum"; do
  i=$((i+1))
  GEMM_POLICY=$POL rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/pmc$i -o p -- python $R/scratch/gemm_one.py $SH 0 > $O/pmc$i.log 2>&1
done
python $R/tools/pmc_digest.py $O k_gemm > $O/digest.txt 2>&1
cat $O/digest.txt
