#!/bin/bash
# round 5, set V (final state of the pose kernels): the whole GPU suite, the rig and the headline against the library of the commit before
# the prologue changes (libaclhip_base.so), then the evidence of the changed kernels again (tools/profile_round5.sh, pose workloads) and the default run
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r5v
(timeout 600 python -m pytest tests -m gpu -x -q > gpurun_out/r5v/gputests.log 2>&1; echo rc=$? >> gpurun_out/r5v/gputests.log)
L=$PWD/acl_amd/lib
V="base:ACLHIP_LIBRARY=$L/libaclhip_base.so new base2:ACLHIP_LIBRARY=$L/libaclhip_base.so new2 base3:ACLHIP_LIBRARY=$L/libaclhip_base.so new3"
python tools/variant_sweep.py --repeats 300 --workloads cinematic,one_clip $V | tee gpurun_out/r5v/sweep.txt
bash tools/profile_round5.sh r05 one_clip one_clip_mixed_registry one_clip_lods 256_clips 256_clips_locality one_clip_qv32 one_clip_qvv40 > gpurun_out/r5v/profile_round.log 2>&1
cd "$GRAFT_REPO_ROOT"
( time python bench.py ) > gpurun_out/r05_bench.json 2> gpurun_out/r05_bench.err
tail -n 3 gpurun_out/r5v/gputests.log
