#!/bin/bash
# A/B of library builds on one box: scratch/lab/lib_*.so are swapped in for the shipped library (on the GPU box's copy only)
L=proxytransformation_amd/libproxyt_hip.so
cp $L /tmp/real.so
for v in "$@"; do
  if [ $v = real ]; then cp /tmp/real.so $L; else cp scratch/lab/lib_$v.so $L; fi
  echo "== $v"
  timeout 120 python scratch/attn_one.py 4 256 196 1 2000
  timeout 120 python scratch/attn_one.py 32 256 196 1 1000
  timeout 120 python scratch/attn_one.py 4 691 196 1 1000
done 2>&1 | grep -v amdgpu.ids
cp /tmp/real.so $L
