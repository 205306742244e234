#!/bin/bash
# Run ON THE GPU BOX (through gpurun): produces the rocprofv3 evidence that profiles/ keeps.
#   tools/profile_round.sh r02
# Writes gpurun_out/<tag>/{bench.json, bench_f32.json, stats_*/, pmc_fetch_*/, pmc_write_*/} and the
# digests <tag>_kernel_stats[_f32].csv / <tag>_pmc_traffic.json next to them.  Counter passes are
# separate runs with --kernel-trace only, FETCH_SIZE and WRITE_SIZE in different passes
# (MI355X_MICROARCH.md: TCC slot budget; no trace domains mixed with --pmc).
set -u
TAG=${1:-r06}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/$TAG
mkdir -p "$O"
export TMPDIR=/tmp
cd /tmp
BENCH="python $R/bench.py --steps 100 --warmup 10 --no-passes"
# 1. the bench lines themselves (default = bf16-stored features; then fp32 features), with the CPU baseline once
python $R/bench.py > "$O/bench.json" 2> "$O/bench.err"
python $R/bench.py --img-dtype f32 --no-cpu-baseline --no-passes > "$O/bench_f32.json" 2>> "$O/bench.err"
# 2. kernel-trace + stats of the same command
for dt in bf16 f32; do
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/stats_$dt" -o k -- \
      $BENCH --no-cpu-baseline --img-dtype $dt > "$O/stats_$dt.log" 2>&1
done
# 3. HBM traffic counters, one counter per pass
for dt in bf16 f32; do
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$O/pmc_${c}_$dt" -o p -- \
        python $R/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-passes --img-dtype $dt > "$O/pmc_${c}_$dt.log" 2>&1
  done
done
# 4. matrix-pipe occupancy of the fp32-equivalent kernels (k_gemm64x / k_gemm128x / k_proxy_attn / k_mlp) where they are large: 32 scenes per GPU
timeout 600 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --output-format csv -d "$O/pmc_mfma_b32" -o p -- \
    python $R/bench.py --scenes-per-gpu 32 --steps 4 --warmup 2 --repeats 1 --setup-forwards 6 --no-cpu-baseline --no-passes > "$O/pmc_mfma_b32.log" 2>&1
python $R/tools/profile_digest.py "$O" "$TAG"
# the headline line once more, now that the traffic digest of THIS build exists (bench.py reads profiles/<tag>_pmc_traffic.json)
cp "$O/${TAG}_pmc_traffic.json" "$R/profiles/${TAG}_pmc_traffic.json"
python $R/bench.py > "$O/bench.json" 2>> "$O/bench.err"
python $R/bench.py --img-dtype f32 --no-cpu-baseline --no-passes > "$O/bench_f32.json" 2>> "$O/bench.err"
for dt in bf16 f32; do
  python $R/tools/timeline.py "$(find "$O/stats_$dt" -name '*kernel_trace.csv' | head -1)" > "$O/${TAG}_timeline_$dt.txt" 2>/dev/null
done
ls "$O"
