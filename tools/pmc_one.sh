#!/bin/bash
# Measurement aid: ONE rocprofv3 counter pass over a bench.py workload. usage: tools/pmc_one.sh <workload> "<counters>" [extra bench.py arguments]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf /tmp/pmc_one
ACLHIP_BENCH_PROFILING=1 timeout 150 rocprofv3 --pmc $2 --output-format csv -d /tmp/pmc_one -o pass -- python bench.py --workload $1 --steps 10 --warmup 2 --no-cpu-baseline ${3:-} > /tmp/pmc_one.log 2>&1
csv=$(find /tmp/pmc_one -name "*counter_collection.csv" | head -1)
python tools/pmc_summary.py decompress $csv | sed "s#^.*csv: ##"
