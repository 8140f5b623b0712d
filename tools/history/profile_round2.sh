#!/bin/bash
# Evidence for profiles/ (round 2): for every north-star workload the bench line, the rocprofv3 --kernel-trace --stats summary of the same
# command, and FETCH_SIZE / WRITE_SIZE in their own --pmc passes (-> traffic.json); then the default bench.py run (all workloads, footprint
# sweep, layouts, CPU thread sweep). usage (on the GPU box): tools/profile_round2.sh <tag> [workload file names ...]   -> gpurun_out/<tag>_*
# (with names, e.g. "scalar object_space": only those workloads are re-measured, the other lines and the default run are left alone)
set -u
tag=${1:-r02}
shift
only="$*"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
rm -f gpurun_out/traffic.json
cp profiles/traffic.json gpurun_out/traffic.json 2>/dev/null
lines=gpurun_out/${tag}_workloads_bench.jsonl
: > $lines
run_workload() {      # <name for files> <traffic key> <bench arguments...>
  name=$1; key=$2; shift 2
  if [ -n "$only" ] && ! echo " $only " | grep -q " $name "; then return; fi
  python bench.py --no-cpu-baseline --no-extras "$@" 2> /dev/null | tail -1 >> $lines
  rm -rf /tmp/prof_trace
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_trace -o trace -- python bench.py --no-cpu-baseline --no-extras "$@" > /dev/null 2> /tmp/trace.log
  stats=$(find /tmp/prof_trace -name "*kernel_stats.csv" | head -1)
  [ -n "$stats" ] && cp "$stats" gpurun_out/${tag}_${name}_kernel_stats.csv
  for counter in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/prof_$counter
    ACLHIP_BENCH_PROFILING=1 timeout 200 rocprofv3 --pmc $counter --output-format csv -d /tmp/prof_$counter -o pass -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras "$@" > /dev/null 2> /tmp/$counter.log
    csv=$(find /tmp/prof_$counter -name "*counter_collection.csv" | head -1)
    [ -n "$csv" ] && cp "$csv" gpurun_out/${tag}_${name}_${counter}.csv
  done
  python tools/traffic_from_pmc.py "$key" gpurun_out/${tag}_${name}_FETCH_SIZE.csv gpurun_out/${tag}_${name}_WRITE_SIZE.csv gpurun_out/traffic.json > /dev/null
  python tools/pmc_summary.py "${PMC_KERNEL_FILTER:-decompress}" gpurun_out/${tag}_${name}_FETCH_SIZE.csv gpurun_out/${tag}_${name}_WRITE_SIZE.csv | sed "s#^.*csv: ##" > gpurun_out/${tag}_${name}_pmc_hbm.txt
  rm -f gpurun_out/${tag}_${name}_FETCH_SIZE.csv gpurun_out/${tag}_${name}_WRITE_SIZE.csv
}
run_workload one_clip "one_clip" --workload one_clip
run_workload 256_clips "256_clips" --workload 256_clips
run_workload 256_clips_locality "256_clips, locality order" --workload 256_clips --order locality
run_workload cinematic "cinematic" --workload cinematic
run_workload database "database" --workload database
run_workload one_clip_qv32 "one_clip, qv32" --workload one_clip --layout qv32
run_workload one_clip_qvv40 "one_clip, qvv40" --workload one_clip --layout qvv40
run_workload scalar "scalar" --workload scalar
run_workload object_space "object_space" --workload object_space
run_workload additive_object_space "additive_object_space" --workload additive_object_space
cp gpurun_out/traffic.json profiles/traffic.json      # the default run below reads the traffic it reports from profiles/traffic.json
if [ -n "$only" ]; then cat $lines; exit 0; fi
( time python bench.py ) > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
python - <<PY
import json
for line in open('$lines'):
    d = json.loads(line)
    print(d['config']['workload'][:60], '|', d['config']['layout'], d['roofline']['kernel'], round(d['roofline']['kernel_ms'] * 1000, 2), 'us', round(d['roofline']['frac'], 3), 'traffic', d['roofline']['traffic'])
d = json.loads(open('gpurun_out/${tag}_bench.json').read().strip().splitlines()[0])
print('default run:', round(d['value'] / 1e9, 3), 'G poses/s', round(d['roofline']['kernel_ms'] * 1000, 2), 'us frac', round(d['roofline']['frac'], 3), 'traffic', d['roofline']['traffic'])
for w in d.get('workloads', []): print(' ', w['workload'], round(w['kernel_ms'] * 1000, 1), round(w['frac'], 3), w['traffic'])
print(' cpu', {k: d['cpu_baseline'][k] for k in ('value', 'threads_at_best', 'per_thread_1t', 'nproc', 'cgroup_cpu_max', 'gpu_over_cpu')})
PY
head -3 gpurun_out/${tag}_one_clip_kernel_stats.csv
