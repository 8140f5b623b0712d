#!/bin/bash
# round 6 A/B: where did decompress_track_kernel's one-clip time go (64.5 us in round 5, 76.7 us after the format / table changes)?
# every variant is the current source with one change compiled out; r05 = round 5's sources. Also the mutated-clip fuzz with its
# differing cases saved. Output: gpurun_out/r06c/
mkdir -p gpurun_out/r06c
export TRACK_SWEEP_SIZES=4194304
for v in r05 ab_current ab_nostoredw ab_scalarplan ab_both r05 ab_current; do
  [ -f acl_amd/lib/libaclhip_$v.so ] || continue
  ACLHIP_LIBRARY=$PWD/acl_amd/lib/libaclhip_$v.so timeout 300 python tools/track_sweep.py 2>&1 | grep decompress_track | sed "s/^/$v: /" | tee -a gpurun_out/r06c/track_ab.txt
done
mkdir -p gpurun_out/r06c/fuzz
FUZZ_SAVE_DIR=gpurun_out/r06c/fuzz timeout 200 python tools/fuzz_gpu_mutated.py 31 8 2>&1 | tail -16
show() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['roofline']['kernel'], round(d['roofline']['kernel_ms']*1000,2), 'us', round(d['roofline']['frac'],4))"; }
for v in r05 ab_current ab_nostoredw; do
  ACLHIP_LIBRARY=$PWD/acl_amd/lib/libaclhip_$v.so timeout 300 python bench.py --workload cinematic --no-extras --no-cpu-baseline 2>/dev/null | show "cinematic $v"
  ACLHIP_LIBRARY=$PWD/acl_amd/lib/libaclhip_$v.so timeout 300 python bench.py --workload one_clip --no-extras --no-cpu-baseline 2>/dev/null | show "one_clip $v"
done
