cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_gpu_consumers.py tests/test_gpu_lifetime.py -q 2>&1 | grep -E "passed|failed|FAILED|Error|assert" | tail -12
for w in object_space additive_object_space scalar; do python bench.py --no-cpu-baseline --no-extras --workload $w 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$w', round(d['roofline']['kernel_ms']*1000,2), 'us frac', round(d['roofline']['frac'],4))"; done
