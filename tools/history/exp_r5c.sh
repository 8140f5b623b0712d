#!/bin/bash
# round 5, experiment set C (one gpurun call): per instance writer decisions -- parity tests, then the headline / LOD / rig / requests lines
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_instance_writers.py tests/test_gpu_layouts.py tests/test_gpu_instance_lists.py tests/test_gpu_parity.py tests/test_gpu_scalar.py tests/test_gpu_consumers.py -m gpu -x -q > gpurun_out/r5c_pytest.txt 2>&1; tail -30 gpurun_out/r5c_pytest.txt
for w in one_clip one_clip_lods cinematic track_requests scalar object_space; do
  python bench.py --workload $w --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$w', d['roofline']['kernel'], round(d['roofline']['kernel_ms']*1000,2), 'us  frac', round(d['roofline']['frac'],4), 'G poses/s', round(d['value']/1e9,3))"
done | tee gpurun_out/r5c_lines.txt
