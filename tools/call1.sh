cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
( time timeout 900 python -m pytest tests -m gpu -x -q ) > gpurun_out/c1_tests.log 2>&1
tail -15 gpurun_out/c1_tests.log
( time timeout 600 python bench.py ) > gpurun_out/c1_bench.json 2> gpurun_out/c1_bench.err
tail -c 600 gpurun_out/c1_bench.err
python - <<'PY'
import json
try:
    d=json.loads(open('gpurun_out/c1_bench.json').read().strip().splitlines()[0])
    print('headline', d['value'], d['roofline']['kernel_ms'], d['roofline']['frac'])
    for w in d.get('workloads',[]): print(w['workload'], round(w['kernel_ms']*1000,1), round(w['frac'],3))
    for w in d.get('footprint_sweep',[]): print('fp', w['instances'], round(w['kernel_ms']*1000,1), round(w['frac'],3))
    for w in d.get('layouts',[]): print('layout', w['layout'], round(w['kernel_ms']*1000,1), round(w['poses_per_s']/1e9,3), round(w['frac'],3))
    print(d.get('cpu_baseline'))
except Exception as e: print('bench parse failed', e)
PY
bash tools/pmc_run.sh cinematic > gpurun_out/c1_pmc_cinematic.log 2>&1
cp gpurun_out/pmc_cinematic.txt gpurun_out/c1_pmc_cinematic.txt
grep -v "^group" gpurun_out/c1_pmc_cinematic.txt | head -60
