# consumers after the negative scale bookkeeping left the launches that cannot meet one
timeout 900 python -m pytest tests/test_gpu_consumers.py -m gpu -x -q 2>&1 | tail -3
for w in object_space additive_object_space; do
  timeout 300 python bench.py --workload $w --steps 300 --warmup 50 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$w', round(d['ms_per_step']*1000,1), d['roofline']['frac'])"
done
