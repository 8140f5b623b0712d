#!/bin/bash
# (history: this set ran on commit 5d59a74 / its working tree -- the knobs and variant libraries it names are not part of the shipped tree; results: profiles/r05_experiments.md 8)
# round 5, set S: set R with the early requests / the arithmetic only in the one-shot pose kernels (template parameters), against base;
# `records` = the same library with ACLHIP_REGULAR_SEGMENTS=0; `cvec` = the consumers with their per lane policy byte loads
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r5s
L=$PWD/acl_amd/lib
V="base:ACLHIP_LIBRARY=$L/libaclhip_base.so new records:ACLHIP_REGULAR_SEGMENTS=0 base2:ACLHIP_LIBRARY=$L/libaclhip_base.so new2 records2:ACLHIP_REGULAR_SEGMENTS=0"
python tools/variant_sweep.py --repeats 300 --workloads one_clip,one_clip_lods,256_clips,cinematic,database $V | tee gpurun_out/r5s/sweep.txt
python tools/variant_sweep.py --repeats 300 --workloads one_clip --layout qv32 $V | tee -a gpurun_out/r5s/sweep.txt
python tools/variant_sweep.py --repeats 300 --workloads one_clip --layout qvv40 $V | tee -a gpurun_out/r5s/sweep.txt
V="base:ACLHIP_LIBRARY=$L/libaclhip_base.so new cvec:ACLHIP_LIBRARY=$L/libaclhip_cvec.so base2:ACLHIP_LIBRARY=$L/libaclhip_base.so new2 cvec2:ACLHIP_LIBRARY=$L/libaclhip_cvec.so"
python tools/variant_sweep.py --repeats 300 --workloads object_space,additive_object_space,blend_object_space,scalar,track_requests $V | tee -a gpurun_out/r5s/sweep.txt
