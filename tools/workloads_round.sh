#!/bin/bash
# Evidence for profiles/: one bench.py line per workload (with its CPU baseline), then FETCH_SIZE / WRITE_SIZE of the mixed-clip
# workload in their own counter passes. usage (on the GPU box): tools/workloads_round.sh <tag>  -> gpurun_out/<tag>_workloads_bench.jsonl
tag=${1:-r01}
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
out=gpurun_out/${tag}_workloads_bench.jsonl
: > $out
for w in one_clip 256_clips cinematic database scalar object_space additive_object_space; do
  timeout 240 python bench.py --workload $w 2> /dev/null | tail -1 >> $out
done
timeout 120 python bench.py --workload 256_clips --sort-by-clip --no-cpu-baseline 2> /dev/null | tail -1 >> $out
ACLHIP_FORCE_GENERIC_KERNEL=1 timeout 120 python bench.py --no-cpu-baseline 2> /dev/null | tail -1 >> $out
python -c "
import json
for line in open('$out'):
    d = json.loads(line)
    print(d['config']['workload'][:70], '|', d['roofline']['kernel'], round(d['ms_per_step'] * 1000, 2), 'us', round(d['roofline']['frac'], 3), 'cpu', d.get('cpu_baseline', {}).get('value'))
"
for counter in FETCH_SIZE WRITE_SIZE; do
  echo "256_clips $counter:"; bash tools/pmc_one.sh 256_clips $counter | tee gpurun_out/${tag}_256_clips_${counter}.txt
done
