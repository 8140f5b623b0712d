#!/bin/bash
# Round 3, experiment D: the staged kernel
cd "$GRAFT_REPO_ROOT"
export ACLHIP_LIBRARY=${ACLHIP_LIBRARY:-acl_amd/lib/libaclhip_exp.so}   # tools/build_experiments.sh
mkdir -p gpurun_out
{
echo "== correctness"
ACLHIP_STAGED=4 ACLHIP_STAGED_ALWAYS=1 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_rows.py tests/test_gpu_full_size.py tests/test_gpu_database.py tests/test_gpu_lifetime.py -x -q 2>&1 | tail -15
ACLHIP_STAGED=8 ACLHIP_STAGED_ALWAYS=1 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_rows.py tests/test_gpu_full_size.py -x -q 2>&1 | tail -5
echo "== timings"
python tools/variant_sweep.py --workloads cinematic,one_clip,256_clips,database \
  base s4:ACLHIP_STAGED=4,ACLHIP_STAGED_ALWAYS=1 s8:ACLHIP_STAGED=8,ACLHIP_STAGED_ALWAYS=1 base2
python tools/variant_sweep.py --workloads 256_clips --order locality base s4:ACLHIP_STAGED=4,ACLHIP_STAGED_ALWAYS=1 s8:ACLHIP_STAGED=8,ACLHIP_STAGED_ALWAYS=1
} 2>&1 | tee gpurun_out/exp_r3d.log
