cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_gpu_adapter.py tests/test_gpu_database.py tests/test_gpu_cpp_mirror.py -q 2>&1 | grep -E "passed|failed|FAILED|Error|mismatch|request" | tail -12
