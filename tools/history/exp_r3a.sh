#!/bin/bash
# Round 3, experiment A: decode -> store wave hand-off on the 300-bone rig (and on the one-window workloads with ACLHIP_HANDOFF_ALWAYS=1)
cd "$GRAFT_REPO_ROOT"
export ACLHIP_LIBRARY=${ACLHIP_LIBRARY:-acl_amd/lib/libaclhip_exp.so}   # tools/build_experiments.sh
mkdir -p gpurun_out
{
echo "== correctness under the hand-off kernels"
ACLHIP_HANDOFF_DECODERS=7 timeout 600 python -m pytest tests/test_gpu_full_size.py -x -q -k "300_bone or hip_graph" 2>&1 | tail -3
ACLHIP_HANDOFF_DECODERS=4 ACLHIP_HANDOFF_LAST_ARRIVER=1 timeout 600 python -m pytest tests/test_gpu_full_size.py -x -q -k "300_bone" 2>&1 | tail -3
ACLHIP_HANDOFF_DECODERS=7 ACLHIP_HANDOFF_ALWAYS=1 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_rows.py -x -q 2>&1 | tail -3
ACLHIP_HANDOFF_DECODERS=3 ACLHIP_HANDOFF_LAST_ARRIVER=1 ACLHIP_HANDOFF_ALWAYS=1 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_rows.py -x -q 2>&1 | tail -3
echo "== timings"
python tools/variant_sweep.py --workloads cinematic \
  base base2 \
  h3:ACLHIP_HANDOFF_DECODERS=3 h4:ACLHIP_HANDOFF_DECODERS=4 h6:ACLHIP_HANDOFF_DECODERS=6 h7:ACLHIP_HANDOFF_DECODERS=7 \
  h8:ACLHIP_HANDOFF_DECODERS=8 h15:ACLHIP_HANDOFF_DECODERS=15 \
  la2:ACLHIP_HANDOFF_DECODERS=2,ACLHIP_HANDOFF_LAST_ARRIVER=1 la3:ACLHIP_HANDOFF_DECODERS=3,ACLHIP_HANDOFF_LAST_ARRIVER=1 \
  la4:ACLHIP_HANDOFF_DECODERS=4,ACLHIP_HANDOFF_LAST_ARRIVER=1 la6:ACLHIP_HANDOFF_DECODERS=6,ACLHIP_HANDOFF_LAST_ARRIVER=1 \
  la8:ACLHIP_HANDOFF_DECODERS=8,ACLHIP_HANDOFF_LAST_ARRIVER=1 base3
python tools/variant_sweep.py --workloads one_clip,256_clips,database \
  base \
  h7:ACLHIP_HANDOFF_DECODERS=7,ACLHIP_HANDOFF_ALWAYS=1 h15:ACLHIP_HANDOFF_DECODERS=15,ACLHIP_HANDOFF_ALWAYS=1 \
  la4:ACLHIP_HANDOFF_DECODERS=4,ACLHIP_HANDOFF_LAST_ARRIVER=1,ACLHIP_HANDOFF_ALWAYS=1 la8:ACLHIP_HANDOFF_DECODERS=8,ACLHIP_HANDOFF_LAST_ARRIVER=1,ACLHIP_HANDOFF_ALWAYS=1
} 2>&1 | tee gpurun_out/exp_r3a.log
