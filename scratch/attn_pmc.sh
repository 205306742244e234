#!/bin/bash
# PMC passes over the fused attention kernel alone (run on the GPU box): scratch/attn_pmc.sh <tag>
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/$1; mkdir -p $O; export TMPDIR=/tmp; cd /tmp
for sh in "4 256 196" "4 256 64" "32 256 196"; do python $R/scratch/attn_one.py $sh 1 50; python $R/scratch/attn_one.py $sh 2 50; done > $O/times.txt 2>&1
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVES" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_ACTIVE_INST_MISC SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/pmc$i -o p -- python $R/scratch/attn_one.py 4 256 196 1 10 > $O/pmc$i.log 2>&1
done
python $R/tools/pmc_digest.py $O k_proxy_attn > $O/digest.txt 2>&1
cat $O/times.txt; cat $O/digest.txt
