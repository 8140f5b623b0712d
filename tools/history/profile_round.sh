#!/bin/bash
# Evidence for profiles/: rocprofv3 kernel-trace stats of the default bench.py run, then FETCH_SIZE / WRITE_SIZE in their own passes.
# usage (on the GPU box): tools/profile_round.sh <tag>      -> gpurun_out/<tag>_*
set -u
tag=${1:-r01}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
python bench.py --no-cpu-baseline > gpurun_out/${tag}_bench_plain.json 2> /dev/null
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_trace -o trace -- python bench.py --no-cpu-baseline > gpurun_out/${tag}_bench_under_trace.json 2> /tmp/trace.log
stats=$(find /tmp/prof_trace -name "*kernel_stats.csv" | head -1)
[ -n "$stats" ] && cp "$stats" gpurun_out/${tag}_one_clip_kernel_stats.csv
for counter in FETCH_SIZE WRITE_SIZE; do
  ACLHIP_BENCH_PROFILING=1 timeout 200 rocprofv3 --pmc $counter --output-format csv -d /tmp/prof_$counter -o pass -- python bench.py --steps 30 --warmup 5 --no-cpu-baseline > /dev/null 2> /tmp/$counter.log
  csv=$(find /tmp/prof_$counter -name "*counter_collection.csv" | head -1)
  [ -n "$csv" ] && cp "$csv" gpurun_out/${tag}_${counter}_counter_collection.csv
done
python tools/traffic_from_pmc.py one_clip gpurun_out/${tag}_FETCH_SIZE_counter_collection.csv gpurun_out/${tag}_WRITE_SIZE_counter_collection.csv gpurun_out/traffic.json
python tools/pmc_summary.py decompress_tracks gpurun_out/${tag}_FETCH_SIZE_counter_collection.csv gpurun_out/${tag}_WRITE_SIZE_counter_collection.csv > gpurun_out/${tag}_one_clip_pmc_hbm.txt
cat gpurun_out/${tag}_bench_plain.json; head -5 gpurun_out/${tag}_one_clip_kernel_stats.csv; cat gpurun_out/${tag}_one_clip_pmc_hbm.txt
