#!/bin/bash
# round 6: the shipped in-turn kernels (arguments from the kernarg segment; exact kernel at 8 waves per SIMD): GPU suite, the rig's evidence files again. Output: gpurun_out/r06n/, gpurun_out/r06_cinematic*
out=gpurun_out/r06n
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 1200 python -m pytest tests -m gpu -x -q -n 2 2>&1 | tail -4 | tee $out/gpu_suite.txt
bash tools/profile_round6.sh r06 cinematic cinematic_fast 2>&1 | tail -4
