python -m pytest tests -m gpu -x -q 2>&1 | tail -1
for v in 0 1 0 1; do echo "rotate $v"; PTX_GEMM_ROTATE=$v python bench.py --steps 50 --warmup 10 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; done
PTX_GEMM_ROTATE=1 python scratch/gemm_time.py 2>&1 | tail -10 | head -6
