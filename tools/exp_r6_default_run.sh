#!/bin/bash
# round 6: on ONE box, the headline workload's evidence files (bench line, rocprofv3 --kernel-trace --stats, HBM and SQ counters), the two mixed-clip track request
# lines, and the default bench run of the final build (headline -> gpurun_out/r06_bench.json, full record -> r06_bench_details.json)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
bash tools/profile_round6.sh r06 one_clip track_requests_256_clips track_requests_256_clips_by_clip 2>&1 | tail -4 | cut -c1-300
( time python bench.py ) > gpurun_out/r06_bench.json 2> gpurun_out/r06_bench.err
cp bench_details.json gpurun_out/r06_bench_details.json
tail -1 gpurun_out/r06_bench.json | wc -c
tail -1 gpurun_out/r06_bench.json | cut -c1-300
tail -4 gpurun_out/r06_bench.err | cut -c1-200
