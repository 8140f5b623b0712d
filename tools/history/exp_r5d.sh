#!/bin/bash
# round 5, experiment set D (one gpurun call): one-window poses taken in turn (2 / 3 / 4 items per wave, image kept while the clip stays) on the
# headline batch, its LOD form and the 256-clip batch; lab build of the library (the knob is a lab knob)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_instance_writers.py -m gpu -x -q > gpurun_out/r5d_pytest.txt 2>&1; tail -5 gpurun_out/r5d_pytest.txt
LAB=ACLHIP_LIBRARY=$PWD/acl_amd/lib/libaclhip_lab.so
python tools/variant_sweep.py --repeats 300 --workloads one_clip,one_clip_lods,256_clips oneshot:$LAB k2:$LAB,ACLHIP_ONE_WINDOW_IN_TURN_ITEMS=2 k3:$LAB,ACLHIP_ONE_WINDOW_IN_TURN_ITEMS=3 k4:$LAB,ACLHIP_ONE_WINDOW_IN_TURN_ITEMS=4 oneshot2:$LAB | tee gpurun_out/r5d_one_window_in_turn.txt
python tools/variant_sweep.py --repeats 300 --workloads 256_clips --order locality oneshot:$LAB k2:$LAB,ACLHIP_ONE_WINDOW_IN_TURN_ITEMS=2 k4:$LAB,ACLHIP_ONE_WINDOW_IN_TURN_ITEMS=4 | tee -a gpurun_out/r5d_one_window_in_turn.txt
