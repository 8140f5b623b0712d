#!/bin/bash
# round 5, experiment set E (one gpurun call): instance lists attached to the caller's clip array -- parity, then the 256-clip frame loop
# as drawn / locality / list (update launch) / attached (caller's kernel), and the headline for regressions
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_instance_lists.py tests/test_gpu_instance_writers.py tests/test_gpu_order_device.py -m gpu -x -q > gpurun_out/r5e_pytest.txt 2>&1; tail -5 gpurun_out/r5e_pytest.txt
python tools/variant_sweep.py --repeats 300 --workloads one_clip,256_clips base | tee gpurun_out/r5e_lists.txt
for order in locality list attached; do python tools/variant_sweep.py --repeats 300 --workloads 256_clips,database --order $order $order | tee -a gpurun_out/r5e_lists.txt; done
python - <<'PY' | tee -a gpurun_out/r5e_lists.txt
import json
for line in open('gpurun_out/variant_sweep.jsonl'):
    d = json.loads(line)
    if d.get('order') == 'attached':
        print('attached', d['workload'], 'step', round(d['kernel_ms']*1000,2), 'us; the caller kernel alone', round(d['caller_update_ms']*1000,2), 'us; orderings', d['list_orderings'])
PY
