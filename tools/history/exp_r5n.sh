#!/bin/bash
# round 5, set N: the pose stores' cache modifiers on the latency-bound layouts (plain / sc1 / nt / sc0 sc1 against the shipped sc0 sc1 nt)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
V=""
for name in plain sc1 nt sc0sc1; do V="$V $name:ACLHIP_LIBRARY=$PWD/acl_amd/lib/libaclhip_st_$name.so"; done
python tools/variant_sweep.py --repeats 300 --workloads one_clip,one_clip_lods,256_clips,cinematic shipped $V shipped2 | tee gpurun_out/r5n_store_modifiers.txt
python tools/variant_sweep.py --repeats 300 --workloads one_clip --layout qv32 shipped $V shipped2 | tee -a gpurun_out/r5n_store_modifiers.txt
python tools/variant_sweep.py --repeats 300 --workloads one_clip --layout qvv40 shipped $V | tee -a gpurun_out/r5n_store_modifiers.txt
python tools/variant_sweep.py --repeats 300 --workloads 256_clips --order locality shipped $V | tee -a gpurun_out/r5n_store_modifiers.txt
