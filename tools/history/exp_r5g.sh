#!/bin/bash
# round 5, set G (one gpurun call): the whole GPU suite, then the default bench.py run (self check, paging on a second stream, VALU floors)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r5g_pytest.txt 2>&1; tail -6 gpurun_out/r5g_pytest.txt
( time python bench.py ) > gpurun_out/r5g_bench.json 2> gpurun_out/r5g_bench.err; tail -5 gpurun_out/r5g_bench.err
python - <<'PY'
import json
d = json.loads(open('gpurun_out/r5g_bench.json').read().strip().splitlines()[-1])
print('default run:', round(d['value'] / 1e9, 3), 'G poses/s', round(d['roofline']['kernel_ms'] * 1000, 2), 'us frac', round(d['roofline']['frac'], 3), 'bound', d['roofline'].get('bound'), 'valu floor', d['roofline'].get('valu_issue_floor_ms'), 'frac_of_bound', d['roofline'].get('frac_of_bound'))
print('self_check', d.get('self_check'))
for w in d.get('workloads', []):
    print(' ', w['workload'], round(w['kernel_ms'] * 1000, 1), 'us', round(w['frac'], 3), 'bound', w.get('bound'), 'valu floor us', None if not w.get('valu_issue_floor_ms') else round(w['valu_issue_floor_ms'] * 1000, 1), 'frac_of_bound', None if w.get('frac_of_bound') is None else round(w['frac_of_bound'], 3), 'traffic', w.get('traffic'))
print([w for w in d['workloads'] if 'second stream' in w['workload']])
print(' cpu', {k: d['cpu_baseline'].get(k) for k in ('value', 'threads_at_best', 'per_thread_1t', 'cold_cache_1t', 'extrapolated_all_cpus', 'extrapolated_physical_cores', 'physical_cores', 'nproc', 'cgroup_cpu_max', 'gpu_over_cpu', 'gpu_over_cpu_extrapolated')})
PY
