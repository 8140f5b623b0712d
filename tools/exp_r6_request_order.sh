#!/bin/bash
# round 6: aclhip_order_track_requests_for_locality -- the new tests, every request pattern, the two mixed-clip lines' evidence files and the default run. Output: gpurun_out/r06s/, gpurun_out/r06_*
out=gpurun_out/r06s
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 1200 python -m pytest tests -m gpu -x -q -n 2 2>&1 | tail -4 | tee $out/gpu_suite.txt
TRACK_SWEEP_SIZES=4194304 timeout 300 python tools/track_sweep.py 2>&1 | grep decompress_track | tee $out/track_patterns.txt
bash tools/profile_round6.sh r06 one_clip track_requests_256_clips track_requests_256_clips_locality 2>&1 | tail -4 | cut -c1-300
( time python bench.py ) > gpurun_out/r06_bench.json 2> gpurun_out/r06_bench.err
cp bench_details.json gpurun_out/r06_bench_details.json
tail -1 gpurun_out/r06_bench.json | wc -c
