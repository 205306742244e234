#!/bin/bash
# whole-step A/B of the GEMM tile policy (run on the GPU box): scratch/r06_gemm_policy_ab.sh <tag>
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/$1; mkdir -p $O; cd $R
for rep in 1 2; do
for pol in 0 256 1; do
  timeout 300 python bench.py --gemm-policy $pol --no-cpu-baseline > $O/cfg2_pol${pol}_$rep.json 2> $O/cfg2_pol${pol}_$rep.err
  timeout 600 python bench.py --gemm-policy $pol --config cfg5 --scenes-per-gpu 16 --steps 10 --warmup 3 --repeats 3 --setup-forwards 6 --no-cpu-baseline --no-passes > $O/cfg5_pol${pol}_$rep.json 2> $O/cfg5_pol${pol}_$rep.err
  timeout 300 python bench.py --gemm-policy $pol --config cfg4_room --scenes-per-gpu 6 --no-cpu-baseline --no-passes > $O/cfg4_pol${pol}_$rep.json 2> $O/cfg4_pol${pol}_$rep.err
done; done
python $R/scratch/r06_policy_digest.py $O
