#!/bin/bash
O=gpurun_out/r05_chunk_overlap2.txt; : > $O
for T in 32 16 8; do
python scratch/img_chunk_overlap2.py $T >> $O 2>&1
for C in 8 4 2 1; do for S in 1 2 3; do
  [ $C -lt $T ] && PTX_LAB_IMG_CHUNK=$C PTX_LAB_IMG_STREAMS=$S python scratch/img_chunk_overlap2.py $T >> $O 2>&1
done; done; done
grep -v amdgpu $O
