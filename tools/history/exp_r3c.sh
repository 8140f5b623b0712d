#!/bin/bash
# Round 3, experiment C: several work items per wave, in turn
cd "$GRAFT_REPO_ROOT"
export ACLHIP_LIBRARY=${ACLHIP_LIBRARY:-acl_amd/lib/libaclhip_exp.so}   # tools/build_experiments.sh
mkdir -p gpurun_out
{
echo "== correctness"
ACLHIP_ITEMS_PER_WAVE=3 timeout 600 python -m pytest tests/test_gpu_full_size.py -x -q -k "300_bone or hip_graph" 2>&1 | tail -3
ACLHIP_ITEMS_PER_WAVE=4 ACLHIP_ITEMS_PER_WAVE_ALWAYS=1 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_rows.py tests/test_gpu_full_size.py -x -q 2>&1 | tail -3
echo "== timings"
python tools/variant_sweep.py --workloads cinematic,one_clip,256_clips,database \
  base k2:ACLHIP_ITEMS_PER_WAVE=2,ACLHIP_ITEMS_PER_WAVE_ALWAYS=1 k3:ACLHIP_ITEMS_PER_WAVE=3,ACLHIP_ITEMS_PER_WAVE_ALWAYS=1 k4:ACLHIP_ITEMS_PER_WAVE=4,ACLHIP_ITEMS_PER_WAVE_ALWAYS=1 \
  k6:ACLHIP_ITEMS_PER_WAVE=6,ACLHIP_ITEMS_PER_WAVE_ALWAYS=1 k8:ACLHIP_ITEMS_PER_WAVE=8,ACLHIP_ITEMS_PER_WAVE_ALWAYS=1 k12:ACLHIP_ITEMS_PER_WAVE=12,ACLHIP_ITEMS_PER_WAVE_ALWAYS=1 base2
} 2>&1 | tee gpurun_out/exp_r3c.log
