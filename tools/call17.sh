cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_gpu_adapter.py -q -x 2>&1 | grep -E "mismatch|float [0-9]|passed|failed" | head -20
