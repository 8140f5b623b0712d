#!/bin/bash
# round 6 A/B: the kernels that take items in turn read their arguments from the kernarg segment per item (0 spilled SGPRs, 65 VGPRs) against the build before (22 spilled, 72),
# and the same at 8 waves per SIMD. Output: gpurun_out/r06l/
out=gpurun_out/r06l
mkdir -p $out
timeout 1200 python -m pytest tests -m gpu -x -q -n 2 2>&1 | tail -4 | tee $out/gpu_suite.txt
show() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['roofline']['kernel'], round(d['roofline']['kernel_ms']*1000,2), 'us', round(d['roofline']['frac'],4))"; }
for round in 1 2 3; do
  for v in ab_prev current ab_turn8; do
    lib=$PWD/acl_amd/lib/libaclhip_$v.so; [ $v = current ] && lib=$PWD/acl_amd/lib/libaclhip.so
    ACLHIP_LIBRARY=$lib timeout 300 python bench.py --workload cinematic --no-extras --no-cpu-baseline 2>/dev/null | show "cinematic $v" | tee -a $out/bench_ab.txt
    ACLHIP_LIBRARY=$lib timeout 300 python bench.py --workload cinematic --fast --no-extras --no-cpu-baseline 2>/dev/null | show "cinematic fast $v" | tee -a $out/bench_ab.txt
  done
done
for v in ab_prev current; do
  lib=$PWD/acl_amd/lib/libaclhip_$v.so; [ $v = current ] && lib=$PWD/acl_amd/lib/libaclhip.so
  ACLHIP_LIBRARY=$lib timeout 300 python bench.py --workload one_clip --no-extras --no-cpu-baseline 2>/dev/null | show "one_clip $v" | tee -a $out/bench_ab.txt
done
