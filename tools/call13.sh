cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocm-smi --showclocks --showpower 2>&1 | head -30
python tools/clock_watch.py cinematic 5 2>&1 | tail -25
