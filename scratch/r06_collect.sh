#!/bin/bash
# copy what profiles/ keeps out of gpurun_out/ (run here, after scratch/r06_final.sh on the GPU box)
cd /root/repo; O=gpurun_out/r06; F=gpurun_out/r06_final; P=profiles
cp $O/bench.json $P/r06_final_bench.json; cp $O/bench_f32.json $P/r06_final_bench_f32.json
cp $O/r06_kernel_stats_bf16.csv $P/r06_final_kernel_stats_bf16.csv; cp $O/r06_kernel_stats_f32.csv $P/r06_final_kernel_stats_f32.csv
cp $O/r06_pmc_traffic.json $P/r06_pmc_traffic.json
cp $O/r06_timeline_bf16.txt $P/r06_final_timeline_bf16.txt; cp $O/r06_timeline_f32.txt $P/r06_final_timeline_f32.txt
for n in b1 b32 cfg1 cfg4 cfg4_b6 cfg4_room_b6 cfg5 cfg5_b16 driver_args pipeline; do cp $F/bench_$n.json $P/r06_final_bench_$n.json; done
cp $F/r06_timeline_cfg4_b6.txt $P/r06_final_timeline_cfg4_b6.txt
grep -v amdgpu.ids $F/vox_time.txt > $P/r06_final_vox_time.txt; grep -v amdgpu.ids $F/ingest_time.txt > $P/r06_final_ingest_time.txt
grep -v amdgpu.ids $F/train_cstep_ab.txt > $P/r06_train_cstep_ab.txt
ls -la $P | grep r06_final | wc -l
