#!/bin/bash
# round 6 measurement: ACLHIP_DECODE_FAST on the rig and on track requests (7 / 6 waves per SIMD for the fast track kernel), the mixed-clip
# track requests after the table pointer moved to ds_bpermute, pipelined device ordering. Output: gpurun_out/r06b/
mkdir -p gpurun_out/r06b
timeout 600 python -m pytest tests/test_gpu_fast_decode.py tests/test_gpu_corpus.py tests/test_gpu_fuzz_slices.py -m gpu -x -q 2>&1 | tail -12
export TRACK_SWEEP_SIZES=4194304
timeout 300 python tools/track_sweep.py > gpurun_out/r06b/track_exact.txt 2>&1
TRACK_SWEEP_FAST=1 timeout 300 python tools/track_sweep.py > gpurun_out/r06b/track_fast7.txt 2>&1
[ -f acl_amd/lib/libaclhip_track6.so ] && ACLHIP_LIBRARY=$PWD/acl_amd/lib/libaclhip_track6.so TRACK_SWEEP_FAST=1 timeout 300 python tools/track_sweep.py > gpurun_out/r06b/track_fast6.txt 2>&1
tail -6 gpurun_out/r06b/track_*.txt
show() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['roofline']['kernel'], round(d['roofline']['kernel_ms']*1000,2), 'us', round(d['roofline']['frac'],4), 'ms/step', round(d['ms_per_step']*1000,2))"; }
for i in 1 2; do
timeout 300 python bench.py --workload cinematic --no-extras --no-cpu-baseline 2>/dev/null | show "cinematic exact"
timeout 300 python bench.py --workload cinematic --fast --no-extras --no-cpu-baseline 2>/dev/null | show "cinematic fast"
done
timeout 300 python bench.py --workload one_clip --fast --no-extras --no-cpu-baseline 2>/dev/null | show "one_clip fast"
timeout 300 python bench.py --workload 256_clips --order device_pipelined --no-extras --no-cpu-baseline 2>/dev/null | show "256 pipelined"
timeout 300 python bench.py --workload 256_clips --order device --no-extras --no-cpu-baseline 2>/dev/null | show "256 device"
timeout 300 python bench.py --workload 256_clips --no-extras --no-cpu-baseline 2>/dev/null | show "256 as drawn"
timeout 300 python bench.py --workload 256_clips --order locality --no-extras --no-cpu-baseline 2>/dev/null | show "256 locality"
