#!/bin/bash
# Measurement aid: bench.py kernel time for the headline workloads, one line each; usage: tools/ab.sh <label> [workloads...]
label=$1; shift
for w in ${@:-one_clip cinematic 256_clips}; do
  python bench.py --workload $w --no-cpu-baseline --no-extras 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$label', '$w', round(d['roofline']['kernel_ms']*1000,2), 'us  frac', round(d['roofline']['frac'],4))"
done
