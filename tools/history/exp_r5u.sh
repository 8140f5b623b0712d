#!/bin/bash
# (history: this set ran on commit 5d59a74 / its working tree -- the knobs and variant libraries it names are not part of the shipped tree; results: profiles/r05_experiments.md 8)
# round 5, set U: the pose consumers without the multi-window loop in the kernel (c1w: a launch of one-window poses never reaches it; its
# presence makes the compiler wait for the base pose DMA before the table rows are requested), with the instance's sample time requested
# next to its clip handle (ct), both (c1wt), against the shipped library; cinematic once more against base
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r5u
L=$PWD/acl_amd/lib
V="new c1w:ACLHIP_LIBRARY=$L/libaclhip_c1w.so ct:ACLHIP_LIBRARY=$L/libaclhip_ct.so c1wt:ACLHIP_LIBRARY=$L/libaclhip_c1wt.so new2 c1w2:ACLHIP_LIBRARY=$L/libaclhip_c1w.so ct2:ACLHIP_LIBRARY=$L/libaclhip_ct.so c1wt2:ACLHIP_LIBRARY=$L/libaclhip_c1wt.so"
python tools/variant_sweep.py --repeats 300 --workloads object_space,object_space_fast,additive_object_space,blend_object_space $V | tee gpurun_out/r5u/sweep.txt
V="base:ACLHIP_LIBRARY=$L/libaclhip_base.so new base2:ACLHIP_LIBRARY=$L/libaclhip_base.so new2 base3:ACLHIP_LIBRARY=$L/libaclhip_base.so new3"
python tools/variant_sweep.py --repeats 300 --workloads cinematic $V | tee -a gpurun_out/r5u/sweep.txt
