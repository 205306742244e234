#!/bin/bash
# Everything profiles/r06_final_* rests on, in one call on the GPU box: scratch/r06_final.sh   (after the last code change)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
export TMPDIR=/tmp
bash tools/profile_round.sh r06 > gpurun_out/profile_round_r06.log 2>&1
O=$R/gpurun_out/r06; F=$R/gpurun_out/r06_final; mkdir -p $F
bash scratch/final_lines.sh r06 > $F/final_lines.log 2>&1
timeout 900 python bench.py --config cfg4_room --scenes-per-gpu 6 --no-cpu-baseline > $F/bench_cfg4_room_b6.json 2> $F/bench_cfg4_room_b6.err
timeout 900 python bench.py --config cfg5 --scenes-per-gpu 16 --steps 10 --warmup 3 --repeats 3 --setup-forwards 6 --no-cpu-baseline > $F/bench_cfg5_b16.json 2> $F/bench_cfg5_b16.err
# BASELINE configs[3] as a chained pipeline (VERDICT r05 "next" #1)
timeout 900 python bench.py --config cfg4_room --pipeline --steps 20 --warmup 5 --repeats 5 > $F/bench_pipeline.json 2> $F/bench_pipeline.err
python scratch/vox_time.py > $F/vox_time.txt 2>&1
python scratch/ingest_time.py 4 > $F/ingest_time.txt 2>&1
timeout 600 python scratch/cluster_regimes.py 4 > $F/cluster_regimes_b4.txt 2>&1
timeout 300 python scratch/train_ab_inproc.py _C_STEP 0 1 > $F/train_cstep_ab.txt 2>&1
ls $O $F
