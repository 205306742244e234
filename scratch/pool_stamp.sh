#!/bin/bash
# on the GPU box: the stamped lab library (python scratch/pool_lab_build.py, here) in place of the shipped one
R=$GRAFT_REPO_ROOT; L=$R/proxytransformation_amd/libproxyt_hip.so; O=$R/gpurun_out; mkdir -p $O
cp $L /tmp/real.so; cp $R/scratch/lab/lib_poolstamp.so $L
for sc in 4 16; do timeout 300 python $R/scratch/pool_stamp.py $sc; done 2>&1 | grep -v amdgpu.ids | tee $O/pool_stamp.txt
cp /tmp/real.so $L
