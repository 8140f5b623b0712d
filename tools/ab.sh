#!/bin/bash
# Measurement aid: runs bench.py (kernel time only) for the three workloads; usage: tools/ab.sh <label>
for w in one_clip cinematic 256_clips; do
  python bench.py --workload $w --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', '$w', round(d['ms_per_step']*1000,2), 'us  frac', round(d['roofline']['frac'],4))"
done
