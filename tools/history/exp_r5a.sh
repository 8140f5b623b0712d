#!/bin/bash
# round 5, experiment set A (one gpurun call): the wave-cooperative decompress_track_kernel -- parity, then its time per request pattern,
# with one aligned 16 byte read per key (shipped) and with the 8 + 4 byte windows (libaclhip_narrow.so: -DACLHIP_TRACK_NARROW_KEYS)
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_random_sweep.py tests/test_gpu_database.py tests/test_gpu_scalar.py -m gpu -x -q 2>&1 | tail -5
echo "== wide"; timeout 300 python tools/track_sweep.py 2>&1 | tee gpurun_out/r5a_track_sweep_wide.txt
echo "== narrow"; ACLHIP_LIBRARY=$PWD/acl_amd/lib/libaclhip_narrow.so timeout 300 python tools/track_sweep.py 2>&1 | tee gpurun_out/r5a_track_sweep_narrow.txt
python bench.py --workload track_requests --no-cpu-baseline --no-extras 2>&1 | tail -1 | tee gpurun_out/r5a_track_bench.json
