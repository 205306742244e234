#!/bin/bash
# Everything profiles/r04_* rests on, in one call on the GPU box: scratch/r04_final.sh   (after the last code change)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
export TMPDIR=/tmp
bash tools/profile_round.sh r04 > gpurun_out/profile_round_r04.log 2>&1
O=$R/gpurun_out/r04; F=$R/gpurun_out/r04_final; mkdir -p $F
bash scratch/final_lines.sh r04 > $F/final_lines.log 2>&1
# cfg5 as the roofline run BASELINE calls it: 16 scenes per GPU, with the per-pass report
timeout 900 python bench.py --config cfg5 --scenes-per-gpu 16 --steps 10 --warmup 3 --repeats 3 --setup-forwards 6 --no-cpu-baseline > $F/bench_cfg5_b16.json 2> $F/bench_cfg5_b16.err
# N2 / N4 stage timings, the training step's kernel statistics, lanes
python scratch/vox_time.py > $F/vox_time.txt 2>&1
python scratch/ingest_time.py 4 > $F/ingest_time.txt 2>&1
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $F/vox -o v -- python $R/scratch/vox_time.py > /dev/null 2>&1 )
grep -i "vox\|fillBuffer" $(find $F/vox -name "*kernel_stats.csv" | head -1) | cut -c1-160 >> $F/vox_time.txt
bash scratch/r04_train_final.sh > $F/train_final.log 2>&1          # -> gpurun_out/r04train/r04_train_{time.txt,kernel_stats.csv,step_trace.txt}
ls $O $F
