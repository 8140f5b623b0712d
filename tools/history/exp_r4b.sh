#!/bin/bash
# round 4, experiment set B (one gpurun call): the shipped in-turn kernel for poses of several windows -- items per wave, the two item
# mappings, on the single-rig batch and on a batch bucketed over several rigs; the consumers with the walk schedule requested early.
cd "$GRAFT_REPO_ROOT"
python tools/variant_sweep.py --repeats 300 --workloads cinematic oneshot:ACLHIP_IN_TURN_ITEMS=1 k2:ACLHIP_IN_TURN_ITEMS=2 k3:ACLHIP_IN_TURN_ITEMS=3 k4:ACLHIP_IN_TURN_ITEMS=4 k5:ACLHIP_IN_TURN_ITEMS=5 \
  k2adj:ACLHIP_IN_TURN_ITEMS=2,ACLHIP_IN_TURN_ADJACENT=1 k4adj:ACLHIP_IN_TURN_ITEMS=4,ACLHIP_IN_TURN_ADJACENT=1 k8adj:ACLHIP_IN_TURN_ITEMS=8,ACLHIP_IN_TURN_ADJACENT=1 oneshot2:ACLHIP_IN_TURN_ITEMS=1
python tools/variant_sweep.py --repeats 200 --workloads object_space,object_space_fast,additive_object_space,additive_object_space_fast,blend_object_space base pad0:ACLHIP_CONSUMER_LDS_PAD=0
python tools/variant_sweep.py --repeats 200 --workloads cinematic_16 oneshot:ACLHIP_IN_TURN_ITEMS=1 k4:ACLHIP_IN_TURN_ITEMS=4 k4adj:ACLHIP_IN_TURN_ITEMS=4,ACLHIP_IN_TURN_ADJACENT=1
python tools/variant_sweep.py --repeats 200 --workloads cinematic_16 --order locality oneshot:ACLHIP_IN_TURN_ITEMS=1 k4:ACLHIP_IN_TURN_ITEMS=4 k4adj:ACLHIP_IN_TURN_ITEMS=4,ACLHIP_IN_TURN_ADJACENT=1
