#!/bin/bash
# round 5, set J: where a light pose's wave spends its life -- phase stamps of the pose kernels (QVV48, QV32, the LOD batch), then the rest of the suite
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export ACLHIP_LIBRARY=$PWD/acl_amd/lib/libaclhip_phase.so
for spec in "one_clip qvv48" "one_clip qv32" "one_clip qvv40" "one_clip_lods qvv48" "256_clips qvv48"; do python tools/phase_times.py $spec; done 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r5j_phase_times.txt
unset ACLHIP_LIBRARY
timeout 900 python -m pytest tests/test_gpu_multiprocess.py tests/test_gpu_fuzz_slices.py tests/test_gpu_all_gather.py -m gpu -x -q > gpurun_out/r5j_pytest.txt 2>&1; grep -n "passed\|failed" gpurun_out/r5j_pytest.txt
