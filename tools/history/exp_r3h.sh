# scalar lists after a change: parity, then the kernel time
timeout 900 python -m pytest tests/test_gpu_scalar.py tests/test_gpu_all_samples.py -m gpu -x -q 2>&1 | tail -3
for i in 1 2; do
timeout 300 python bench.py --workload scalar --steps 600 --warmup 100 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('scalar', round(d['roofline']['kernel_ms']*1000,2), round(d['roofline']['frac'],3), d['roofline']['kernel'])"
done
