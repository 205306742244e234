python -m pytest tests -m gpu -x -q 2>&1 | tail -1
python bench.py --steps 50 --warmup 10 --no-cpu-baseline --breakdown 2>&1 | tail -2 | python -c "
import sys,json
l=sys.stdin.read().strip().split('\n')
bd=json.loads(l[0].split('launch: ')[1]); d=json.loads(l[1])
print(d['value'], d['value_f32_features'], d['roofline']['avg_launch_us'], d['roofline']['frac'], {k:bd[k] for k in bd if 'img' in k})"
python bench.py --steps 50 --warmup 10 --no-cpu-baseline --img-dtype f32 --breakdown 2>&1 | tail -2 | python -c "
import sys,json
l=sys.stdin.read().strip().split('\n')
bd=json.loads(l[0].split('launch: ')[1]); d=json.loads(l[1])
print('f32', d['value'], d['roofline']['avg_launch_us'], d['roofline']['frac'], {k:bd[k] for k in bd if 'img' in k})"
