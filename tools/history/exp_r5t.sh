#!/bin/bash
# round 5, set T: what was kept of sets P .. S (a wave's sample time requested next to its clip handle, the pose kernels' per instance
# policies through the scalar unit, the clip range requested in front of the plan entries) against the library of the commit before
# (libaclhip_base.so), after the whole GPU suite
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r5t
(timeout 600 python -m pytest tests -m gpu -x -q > gpurun_out/r5t/gputests.log 2>&1; echo rc=$? >> gpurun_out/r5t/gputests.log)
L=$PWD/acl_amd/lib
V="base:ACLHIP_LIBRARY=$L/libaclhip_base.so new base2:ACLHIP_LIBRARY=$L/libaclhip_base.so new2 base3:ACLHIP_LIBRARY=$L/libaclhip_base.so new3"
python tools/variant_sweep.py --repeats 300 --workloads one_clip,one_clip_lods,256_clips,cinematic,database,object_space $V | tee gpurun_out/r5t/sweep.txt
python tools/variant_sweep.py --repeats 300 --workloads one_clip --layout qv32 $V | tee -a gpurun_out/r5t/sweep.txt
python tools/variant_sweep.py --repeats 300 --workloads one_clip --layout qvv40 $V | tee -a gpurun_out/r5t/sweep.txt
python tools/variant_sweep.py --repeats 300 --workloads 256_clips,database --order locality $V | tee -a gpurun_out/r5t/sweep.txt
tail -n 3 gpurun_out/r5t/gputests.log
