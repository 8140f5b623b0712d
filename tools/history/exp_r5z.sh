#!/bin/bash
# round 5, set Z: the pose-related GPU tests under the alternate kernel paths, final build (one-shot waves for poses of several windows; 16 byte
# key reads everywhere; the compiler's square roots everywhere; the any-settings kernels for every launch)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r5z
T="tests/test_gpu_parity.py tests/test_gpu_layouts.py tests/test_gpu_launch_shape.py tests/test_gpu_key_frames.py tests/test_gpu_instance_writers.py tests/test_gpu_rows.py tests/test_gpu_full_size.py tests/test_gpu_database.py tests/test_gpu_instance_lists.py tests/test_gpu_lifetime.py"
for v in "ACLHIP_IN_TURN_ITEMS=1" "ACLHIP_WIDE_KEY_LOADS=1" "ACLHIP_SHORT_EXACT_MATH=0" "ACLHIP_FORCE_GENERIC_KERNEL=1"; do
  echo "== $v" | tee -a gpurun_out/r5z/alternate_paths.txt
  (env $v timeout 400 python -m pytest $T -x -q 2>&1 | tail -n 2) | tee -a gpurun_out/r5z/alternate_paths.txt
done
