cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_layouts.py tests/test_gpu_rows.py tests/test_gpu_full_size.py -q -x 2>&1 | tail -5
bash tools/ab.sh looped cinematic one_clip 256_clips database
