#!/bin/bash
# round 6: the default bench run of the final build (headline -> gpurun_out/r06_bench.json, full record -> r06_bench_details.json)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
( time python bench.py ) > gpurun_out/r06_bench.json 2> gpurun_out/r06_bench.err
cp bench_details.json gpurun_out/r06_bench_details.json
tail -1 gpurun_out/r06_bench.json | cut -c1-600
