python -m pytest tests -m gpu -x -q 2>&1 | tail -1
for v in 0 1; do python bench.py --steps 50 --warmup 10 --no-cpu-baseline --time-kernel k_img_gather 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['avg_launch_us'], d['roofline']['frac'])"; done
