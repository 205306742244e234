#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
for B in 4 8 16 32; do for nc in 0 1; do
  if [ $nc = 1 ]; then export PTX_NO_CHAIN=1; else unset PTX_NO_CHAIN; fi
  timeout 200 python bench.py --scenes-per-gpu $B --steps 30 --warmup 6 --no-cpu-baseline --no-passes 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('B=$B no_chain=$nc', d['value'], d['ms_per_step'])"
done; done
