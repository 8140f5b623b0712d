#!/bin/bash
# (history: this set ran on commit 5d59a74 / its working tree -- the knobs and variant libraries it names are not part of the shipped tree; results: profiles/r05_experiments.md 8)
# round 5, set P: a wave's sample time requested next to its clip handle (not behind the clip record), a key's segment by arithmetic
# (table rows requested with the sample records, not behind them) -- against the library of the commit before (libaclhip_base.so,
# built from `git archive 4649d62`) and against ACLHIP_REGULAR_SEGMENTS=0 (every clip through its sample records)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r5p
(timeout 300 python -m pytest tests/test_gpu_segment_map.py tests/test_gpu_parity.py tests/test_gpu_database.py tests/test_gpu_layouts.py -x -q > gpurun_out/r5p/tests.log 2>&1; echo rc=$? >> gpurun_out/r5p/tests.log)
V="base:ACLHIP_LIBRARY=$PWD/acl_amd/lib/libaclhip_base.so records:ACLHIP_REGULAR_SEGMENTS=0 arithmetic base2:ACLHIP_LIBRARY=$PWD/acl_amd/lib/libaclhip_base.so arithmetic2"
python tools/variant_sweep.py --repeats 300 --workloads one_clip,one_clip_lods,256_clips,cinematic,database,track_requests,object_space $V | tee gpurun_out/r5p/sweep.txt
python tools/variant_sweep.py --repeats 300 --workloads one_clip --layout qv32 $V | tee -a gpurun_out/r5p/sweep.txt
python tools/variant_sweep.py --repeats 300 --workloads one_clip --layout qvv40 $V | tee -a gpurun_out/r5p/sweep.txt
python tools/variant_sweep.py --repeats 300 --workloads 256_clips,database --order locality $V | tee -a gpurun_out/r5p/sweep.txt
tail -n 3 gpurun_out/r5p/tests.log
