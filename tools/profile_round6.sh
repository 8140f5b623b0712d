#!/bin/bash
# Evidence for profiles/ (round 6: round 5's recipe + the ACLHIP_DECODE_FAST and pipelined-ordering workloads; the default run's stdout is the < 4 KB headline, its full record goes to bench_details.json): for every workload of the default run the bench line, the rocprofv3 --kernel-trace --stats summary of the
# same command, FETCH_SIZE / WRITE_SIZE in their own --pmc passes (-> traffic.json) and, for the workloads named in SQ_WORKLOADS, the SQ /
# texture unit counters of the shipped kernel; then the default bench.py run (which measures every entry's traffic itself).
# usage (on the GPU box): tools/profile_round6.sh <tag> [workload file names ...]   -> gpurun_out/<tag>_*
set -u
tag=${1:-r06}
shift
only="$*"
SQ_WORKLOADS=" one_clip one_clip_lods 256_clips 256_clips_attached cinematic cinematic_fast database scalar object_space object_space_fast additive_object_space additive_object_space_fast blend_object_space track_requests track_requests_256_clips track_requests_256_clips_locality "
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
  echo "$SQ_WORKLOADS" | grep -q " $name " || return
  # SQ / texture unit counters of the shipped kernel, one group per pass
  sq=gpurun_out/${tag}_${name}_pmc_sq.txt
  : > $sq
  for group in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_INST_ANY" \
               "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_WAIT_ANY" \
               "GRBM_GUI_ACTIVE TA_TA_BUSY_sum TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum TCC_EA0_WRREQ_STALL_sum"; do
    rm -rf /tmp/prof_sq
    ACLHIP_BENCH_PROFILING=1 timeout 200 rocprofv3 --pmc $group --output-format csv -d /tmp/prof_sq -o pass -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras "$@" > /dev/null 2> /tmp/sq.log
    csv=$(find /tmp/prof_sq -name "*counter_collection.csv" | head -1)
    [ -n "$csv" ] && python tools/pmc_summary.py "${PMC_KERNEL_FILTER:-decompress}" $csv | sed "s#^.*csv: ##" >> $sq
  done
}
run_workload one_clip "one_clip" --workload one_clip
run_workload one_clip_mixed_registry "one_clip_mixed_registry" --workload one_clip_mixed_registry
run_workload one_clip_lods "one_clip_lods" --workload one_clip_lods
run_workload 256_clips "256_clips" --workload 256_clips
run_workload 256_clips_locality "256_clips, locality order" --workload 256_clips --order locality
run_workload 256_clips_list "256_clips, list order" --workload 256_clips --order list
run_workload 256_clips_device "256_clips, device order" --workload 256_clips --order device
run_workload 256_clips_attached "256_clips, attached order" --workload 256_clips --order attached
run_workload 256_clips_device_pipelined "256_clips, device_pipelined order" --workload 256_clips --order device_pipelined
run_workload cinematic "cinematic" --workload cinematic
run_workload cinematic_fast "cinematic, fast" --workload cinematic --fast
run_workload database "database" --workload database
run_workload database_locality "database, locality order" --workload database --order locality
run_workload database_list "database, list order" --workload database --order list
run_workload one_clip_qv32 "one_clip, qv32" --workload one_clip --layout qv32
run_workload one_clip_qvv40 "one_clip, qvv40" --workload one_clip --layout qvv40
run_workload track_requests "track_requests" --workload track_requests
run_workload track_requests_256_clips "track_requests_256_clips" --workload track_requests_256_clips
run_workload track_requests_256_clips_locality "track_requests_256_clips, locality order" --workload track_requests_256_clips --order locality
run_workload scalar "scalar" --workload scalar
run_workload object_space "object_space" --workload object_space
run_workload object_space_fast "object_space_fast" --workload object_space_fast
run_workload additive_object_space "additive_object_space" --workload additive_object_space
run_workload additive_object_space_fast "additive_object_space_fast" --workload additive_object_space_fast
run_workload blend_object_space "blend_object_space" --workload blend_object_space
cp gpurun_out/traffic.json profiles/traffic.json      # what a default run falls back to when it cannot measure its traffic itself
if [ -n "$only" ]; then cat $lines; exit 0; fi
( time python bench.py ) > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
cp bench_details.json gpurun_out/${tag}_bench_details.json
python - <<PY
import json
for line in open('$lines'):
    d = json.loads(line)
    print(d['config']['workload'][:60], '|', d['config']['layout'], d['roofline']['kernel'], round(d['roofline']['kernel_ms'] * 1000, 2), 'us', round(d['roofline']['frac'], 3), 'traffic', d['roofline']['traffic'])
text = open('gpurun_out/${tag}_bench.json').read().strip().splitlines()[-1]
d = json.loads(text)
print('default run: headline of', len(text), 'bytes:', round(d['value'] / 1e9, 3), 'G poses/s', round(d['roofline']['kernel_ms'] * 1000, 2), 'us frac', round(d['roofline']['frac'], 3), 'traffic', d['roofline']['traffic'])
print(' self_check', d.get('self_check'))
for name, row in d.get('workloads', {}).items(): print(' ', name, row)
print(' cpu', d.get('cpu_baseline'))
PY
head -3 gpurun_out/${tag}_one_clip_kernel_stats.csv
