cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_layouts.py tests/test_gpu_consumers.py -q -x 2>&1 | tail -3
bash tools/ab.sh late_dma cinematic one_clip 256_clips database
ACLHIP_LIBRARY=$PWD/acl_amd/lib/libaclhip_early.so bash tools/ab.sh early_dma cinematic one_clip 256_clips database
