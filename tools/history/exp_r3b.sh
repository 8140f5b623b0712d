#!/bin/bash
# Round 3, experiment B: persistent decode waves + store waves
cd "$GRAFT_REPO_ROOT"
export ACLHIP_LIBRARY=${ACLHIP_LIBRARY:-acl_amd/lib/libaclhip_exp.so}   # tools/build_experiments.sh
mkdir -p gpurun_out
{
echo "== correctness under the persistent kernels"
ACLHIP_PERSISTENT=71 timeout 600 python -m pytest tests/test_gpu_full_size.py -x -q -k "300_bone or hip_graph" 2>&1 | tail -3
ACLHIP_PERSISTENT=142 ACLHIP_PERSISTENT_ALWAYS=1 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_rows.py tests/test_gpu_full_size.py -x -q 2>&1 | tail -3
ACLHIP_PERSISTENT=31 ACLHIP_PERSISTENT_ALWAYS=1 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_rows.py -x -q 2>&1 | tail -3
echo "== timings"
python tools/variant_sweep.py --workloads cinematic,one_clip,256_clips,database \
  base p31:ACLHIP_PERSISTENT=31,ACLHIP_PERSISTENT_ALWAYS=1 p71:ACLHIP_PERSISTENT=71,ACLHIP_PERSISTENT_ALWAYS=1 p62:ACLHIP_PERSISTENT=62,ACLHIP_PERSISTENT_ALWAYS=1 \
  p142:ACLHIP_PERSISTENT=142,ACLHIP_PERSISTENT_ALWAYS=1 p151:ACLHIP_PERSISTENT=151,ACLHIP_PERSISTENT_ALWAYS=1 base2
} 2>&1 | tee gpurun_out/exp_r3b.log
