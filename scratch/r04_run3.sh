#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
O=$R/gpurun_out/r04d; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_ingest.py tests/test_gpu_workloads.py -m gpu -x -q -k "index_kernel or cfg5 or ingest" > $O/gputest.txt 2>&1; tail -4 $O/gputest.txt
python scratch/ingest_time.py 4 > $O/ingest_time.txt 2>&1; tail -3 $O/ingest_time.txt
timeout 900 python bench.py --config cfg5 --scenes-per-gpu 16 --steps 10 --warmup 3 --repeats 3 --setup-forwards 6 --no-cpu-baseline > $O/bench_cfg5_b16.json 2> $O/bench_cfg5_b16.err; tail -c 3000 $O/bench_cfg5_b16.json; tail -5 $O/bench_cfg5_b16.err
