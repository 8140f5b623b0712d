#!/bin/bash
# round 6, one box, the FINAL build: the in-turn kernels at 6 waves per SIMD against 7 (rig), the GPU suite, then every workload's evidence (tools/profile_round6.sh r06)
out=gpurun_out/r06k
mkdir -p $out
show() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['roofline']['kernel'], round(d['roofline']['kernel_ms']*1000,2), 'us', round(d['roofline']['frac'],4))"; }
for round in 1 2; do
  for v in current ab_turn6; do
    lib=$PWD/acl_amd/lib/libaclhip_$v.so; [ $v = current ] && lib=$PWD/acl_amd/lib/libaclhip.so
    ACLHIP_LIBRARY=$lib timeout 300 python bench.py --workload cinematic --no-extras --no-cpu-baseline 2>/dev/null | show "cinematic $v" | tee -a $out/in_turn_waves.txt
  done
done
timeout 1200 python -m pytest tests -m gpu -x -q -n 2 2>&1 | tail -4 | tee $out/gpu_suite.txt
bash tools/profile_round6.sh r06 2>&1 | tail -60 | tee $out/profile_tail.txt
