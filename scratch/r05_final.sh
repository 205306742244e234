#!/bin/bash
# Everything profiles/r05_final_* rests on, in one call on the GPU box: scratch/r05_final.sh   (after the last code change)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
export TMPDIR=/tmp
bash tools/profile_round.sh r05 > gpurun_out/profile_round_r05.log 2>&1
O=$R/gpurun_out/r05; F=$R/gpurun_out/r05_final; mkdir -p $F
bash scratch/final_lines.sh r05 > $F/final_lines.log 2>&1
# the shipped configuration in the room regime at its training batch (VERDICT r04 +J2), with the per-pass report
timeout 900 python bench.py --config cfg4_room --scenes-per-gpu 6 --no-cpu-baseline > $F/bench_cfg4_room_b6.json 2> $F/bench_cfg4_room_b6.err
# cfg5 as the roofline run BASELINE calls it: 16 scenes per GPU, with the per-pass report
timeout 900 python bench.py --config cfg5 --scenes-per-gpu 16 --steps 10 --warmup 3 --repeats 3 --setup-forwards 6 --no-cpu-baseline > $F/bench_cfg5_b16.json 2> $F/bench_cfg5_b16.err
python scratch/vox_time.py > $F/vox_time.txt 2>&1
python scratch/ingest_time.py 4 > $F/ingest_time.txt 2>&1
ls $O $F
