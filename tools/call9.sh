cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
( time timeout 900 python -m pytest tests -m gpu -q ) > gpurun_out/c9_tests.log 2>&1
tail -12 gpurun_out/c9_tests.log
( time timeout 600 python bench.py --no-cpu-baseline ) > gpurun_out/c9_bench.json 2> gpurun_out/c9_bench.err
tail -c 300 gpurun_out/c9_bench.err
python - <<'PY'
import json
try:
    d=json.loads(open('gpurun_out/c9_bench.json').read().strip().splitlines()[0])
    print('headline', d['value'], d['roofline']['kernel_ms'], d['roofline']['frac'])
    for w in d.get('workloads',[]): print(w['workload'], round(w['kernel_ms']*1000,1), round(w['frac'],3))
    for w in d.get('footprint_sweep',[]): print('fp', w['instances'], round(w['kernel_ms']*1000,1), round(w['frac'],3))
    for w in d.get('layouts',[]): print('layout', w['layout'], round(w['kernel_ms']*1000,1), round(w['poses_per_s']/1e9,3), round(w['frac'],3))
except Exception as e: print('bench parse failed', e)
PY
