cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_gpu_lifetime.py -q -x -s 2>&1 | tail -15
timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -5
