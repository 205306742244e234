# usage: tl.sh TAG [bench args...] -> timeline of one steady-state step under rocprofv3
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; TAG=$1; shift
cd /tmp && rm -rf $R/gpurun_out/tl_$TAG && rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/tl_$TAG -o k -- python $R/bench.py --steps 40 --warmup 10 --no-cpu-baseline "$@" > $R/gpurun_out/tl_$TAG.log 2>&1
python $R/tools/timeline.py $(find $R/gpurun_out/tl_$TAG -name "k_kernel_trace.csv") 30
