#!/bin/bash
# round 6 A/B: mixed-clip track requests with the clip record heads gathered four lanes per record (current) against per lane record loads (ab_prev). Output: gpurun_out/r06h/
mkdir -p gpurun_out/r06h
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_corpus.py tests/test_gpu_database.py tests/test_gpu_fast_decode.py tests/test_gpu_instance_writers.py -m gpu -x -q -n 2 2>&1 | tail -4 | tee gpurun_out/r06h/tests.txt
export TRACK_SWEEP_SIZES=4194304
for v in ab_prev current ab_rec6 ab_quads ab_quads6 ab_prev current; do
  lib=$PWD/acl_amd/lib/libaclhip_$v.so; [ $v = current ] && lib=$PWD/acl_amd/lib/libaclhip.so
  ACLHIP_LIBRARY=$lib timeout 300 python tools/track_sweep.py 2>&1 | grep decompress_track | sed "s/^/$v: /" | tee -a gpurun_out/r06h/track_ab.txt
done
TRACK_SWEEP_FAST=1 timeout 300 python tools/track_sweep.py 2>&1 | grep decompress_track | sed "s/^/current fast: /" | tee -a gpurun_out/r06h/track_ab.txt
show() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['roofline']['kernel'], round(d['roofline']['kernel_ms']*1000,2), 'us', round(d['roofline']['frac'],4))"; }
for v in ab_prev current; do
  lib=$PWD/acl_amd/lib/libaclhip_$v.so; [ $v = current ] && lib=$PWD/acl_amd/lib/libaclhip.so
  for w in one_clip cinematic 256_clips track_requests; do
    ACLHIP_LIBRARY=$lib timeout 300 python bench.py --workload $w --no-extras --no-cpu-baseline 2>/dev/null | show "$w $v" | tee -a gpurun_out/r06h/bench_ab.txt
  done
done
