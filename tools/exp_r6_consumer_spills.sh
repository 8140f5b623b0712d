#!/bin/bash
# round 6 A/B: the pose consumers with every field of their clip records in registers of its own (scalar spills 27 -> 14 in the additive kernel; round 5 had 16) and the
# device_clip layout with the pointers back in the first register block, against the build before (ab_prev). Output: gpurun_out/r06j/
out=gpurun_out/r06j
mkdir -p $out
timeout 1200 python -m pytest tests -m gpu -x -q -n 2 2>&1 | tail -4 | tee $out/gpu_suite.txt
show() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['roofline']['kernel'], round(d['roofline']['kernel_ms']*1000,2), 'us', round(d['roofline']['frac'],4))"; }
for round in 1 2; do
for v in ab_prev current; do
  lib=$PWD/acl_amd/lib/libaclhip_$v.so; [ $v = current ] && lib=$PWD/acl_amd/lib/libaclhip.so
  for w in additive_object_space additive_object_space_fast object_space object_space_fast blend_object_space; do
    ACLHIP_LIBRARY=$lib timeout 300 python bench.py --workload $w --no-extras --no-cpu-baseline 2>/dev/null | show "$w $v" | tee -a $out/bench_ab.txt
  done
done
done
for v in ab_prev current; do
  lib=$PWD/acl_amd/lib/libaclhip_$v.so; [ $v = current ] && lib=$PWD/acl_amd/lib/libaclhip.so
  for w in one_clip cinematic 256_clips track_requests scalar database; do
    ACLHIP_LIBRARY=$lib timeout 300 python bench.py --workload $w --no-extras --no-cpu-baseline 2>/dev/null | show "$w $v" | tee -a $out/bench_ab.txt
  done
done
export TRACK_SWEEP_SIZES=4194304
timeout 300 python tools/track_sweep.py 2>&1 | grep decompress_track | tee $out/track_patterns.txt
