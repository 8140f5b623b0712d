#!/bin/bash
# round 4, experiment set A (one gpurun call): the rig's in-turn kernels with 16 byte key reads at 8 / 7 / 6 waves per SIMD of registers; the
# pose consumers with padded LDS images and with ACLHIP_CONSUMERS_FAST. usage: tools/exp_r4a.sh  -> gpurun_out/exp_r4a.log
cd "$GRAFT_REPO_ROOT"
export ACLHIP_LIBRARY=$PWD/acl_amd/lib/libaclhip_exp.so
python tools/variant_sweep.py --repeats 300 --workloads cinematic base \
  turn4w1:ACLHIP_ITEMS_PER_WAVE=4,ACLHIP_ITEMS_PER_WAVE_WIDE=1 turn8w1:ACLHIP_ITEMS_PER_WAVE=8,ACLHIP_ITEMS_PER_WAVE_WIDE=1 \
  turn4w2:ACLHIP_ITEMS_PER_WAVE=4,ACLHIP_ITEMS_PER_WAVE_WIDE=2 turn6w2:ACLHIP_ITEMS_PER_WAVE=6,ACLHIP_ITEMS_PER_WAVE_WIDE=2 turn8w2:ACLHIP_ITEMS_PER_WAVE=8,ACLHIP_ITEMS_PER_WAVE_WIDE=2 \
  turn6w3:ACLHIP_ITEMS_PER_WAVE=6,ACLHIP_ITEMS_PER_WAVE_WIDE=3 turn8w0:ACLHIP_ITEMS_PER_WAVE=8 base2
unset ACLHIP_LIBRARY
python tools/variant_sweep.py --repeats 200 --workloads object_space,object_space_fast,additive_object_space,additive_object_space_fast base \
  pad16:ACLHIP_CONSUMER_LDS_PAD=16 pad32:ACLHIP_CONSUMER_LDS_PAD=32 pad64:ACLHIP_CONSUMER_LDS_PAD=64 \
  inst8:ACLHIP_CONSUMER_LOG2_INSTANCES=3 inst8pad16:ACLHIP_CONSUMER_LOG2_INSTANCES=3,ACLHIP_CONSUMER_LDS_PAD=16 inst2:ACLHIP_CONSUMER_LOG2_INSTANCES=1
