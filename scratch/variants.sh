python -m pytest tests -m gpu -x -q 2>&1 | tail -1
for v in 0 1; do echo "scores dbg $v"; PTX_SCORES_DBG=$v python bench.py --steps 30 --warmup 5 --no-cpu-baseline --img-dtype f32 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['avg_launch_us'], d['roofline']['frac'])"; done
python bench.py --steps 50 --warmup 10 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bf16:', d['value'], d['value_f32_features'], d['roofline']['avg_launch_us'], d['roofline']['frac'])"
