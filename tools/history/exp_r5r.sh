#!/bin/bash
# (history: this set ran on commit 5d59a74 / its working tree -- the knobs and variant libraries it names are not part of the shipped tree; results: profiles/r05_experiments.md 8)
# round 5, set R: the pose kernels' prologue without its vector memory wait (per instance policies through the scalar unit:
# uniform_instance_byte) + sets P's changes, against the library of the commit before (libaclhip_base.so); then the phase stamps
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r5r
(timeout 300 python -m pytest tests/test_gpu_segment_map.py tests/test_gpu_parity.py tests/test_gpu_database.py tests/test_gpu_layouts.py tests/test_gpu_instance_writers.py tests/test_gpu_scalar.py tests/test_gpu_consumers.py -x -q > gpurun_out/r5r/tests.log 2>&1; echo rc=$? >> gpurun_out/r5r/tests.log)
V="base:ACLHIP_LIBRARY=$PWD/acl_amd/lib/libaclhip_base.so new base2:ACLHIP_LIBRARY=$PWD/acl_amd/lib/libaclhip_base.so new2"
python tools/variant_sweep.py --repeats 300 --workloads one_clip,one_clip_lods,256_clips,cinematic,database,scalar,object_space $V | tee gpurun_out/r5r/sweep.txt
python tools/variant_sweep.py --repeats 300 --workloads one_clip --layout qv32 $V | tee -a gpurun_out/r5r/sweep.txt
python tools/variant_sweep.py --repeats 300 --workloads one_clip --layout qvv40 $V | tee -a gpurun_out/r5r/sweep.txt
python tools/variant_sweep.py --repeats 300 --workloads 256_clips,database --order locality $V | tee -a gpurun_out/r5r/sweep.txt
for lib in phase1 phase2; do
	export ACLHIP_LIBRARY=$PWD/acl_amd/lib/libaclhip_$lib.so
	for spec in "one_clip qvv48" "256_clips qvv48"; do echo "== $lib"; python tools/phase_times.py $spec; done
done 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r5r/phase_times.txt
tail -n 3 gpurun_out/r5r/tests.log
