#!/bin/bash
# r05: eval long run (memory / speed stability) + kernel trace of the training step with the two blocks on two streams
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/apart
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
#timeout 120 python $R/scratch/eval_long_run.py 6000 cfg2 2>&1 | grep -v amdgpu.ids > $O/eval_long_run.txt
for ap in 1 0; do
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/tr$ap -o t -- python $R/scratch/train_long_run.py 300 1 $ap 1 > $O/trace$ap.log 2>&1
  f=$(find $O/tr$ap -name '*kernel_trace.csv' | head -1)
  python $R/scratch/apart_trace_digest.py $f > $O/digest$ap.txt 2>&1
  rm -rf $O/tr$ap
done
tail -5 $O/eval_long_run.txt; cat $O/trace1.log | grep steps; cat $O/digest1.txt | tail -30; cat $O/digest0.txt | tail -12
