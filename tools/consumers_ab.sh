#!/bin/bash
# Measurement aid: the pose consumer workloads for 1 / 2 / 4 / 8 instances per workgroup; usage: tools/consumers_ab.sh [log2 values]
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
for g in ${@:-3 2 1 0}; do
for w in object_space additive_object_space; do
  ACLHIP_CONSUMER_LOG2_INSTANCES=$g timeout 120 python bench.py --workload $w --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('log2instances=$g', '$w', round(d['ms_per_step']*1000,2), 'us  b2b', round(d['roofline']['kernel_ms_back_to_back']*1000,2), 'frac', round(d['roofline']['frac'],4))"
done
done 2>&1 | tee gpurun_out/consumers_bench.log
