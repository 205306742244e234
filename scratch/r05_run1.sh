#!/bin/bash
# r05 first call: the two parity gaps VERDICT r04 names (+J2 room regime, 32-scene call), then the room bench line with its layout A/B
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
export TMPDIR=/tmp
O=$R/gpurun_out/r05a; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_workloads.py -x -q -m gpu -k "room_regime or 32_scenes" > $O/tests.log 2>&1
tail -5 $O/tests.log
B="python bench.py --config cfg4_room --scenes-per-gpu 6 --steps 40 --warmup 8 --repeats 3 --no-cpu-baseline --no-passes"
timeout 600 $B > $O/room_b6.json 2> $O/room_b6.err
for i in 1 2; do
for v in "PTX_CHAIN_SWAP=0" "PTX_CHAIN_SWAP=1" "PTX_EARLY_PROXIES=0" "PTX_EARLY_PROXIES=1" "PTX_IMG_AFTER_CLUSTER=0" "PTX_IMG_AFTER_CLUSTER=1"; do
  echo "== $v" >> $O/room_ab.txt
  env $v timeout 600 $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d.get('timed_blocks'))" >> $O/room_ab.txt
done; done
timeout 600 python bench.py --config cfg4 --scenes-per-gpu 6 --steps 40 --warmup 8 --repeats 3 --no-cpu-baseline --no-passes > $O/cfg4_b6.json 2>/dev/null
cat $O/room_ab.txt
head -c 600 $O/room_b6.json
