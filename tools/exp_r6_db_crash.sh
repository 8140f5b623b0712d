#!/bin/bash
# round 6: the mutated-database fuzz again (bulk buffers as long as the mutated header claims), then the GPU suite. Output: gpurun_out/r06f/
out=gpurun_out/r06f
mkdir -p $out/fuzz
for seed in 71 72 73 74; do
  FUZZ_VERBOSE=1 FUZZ_SAVE_DIR=$out/fuzz timeout 200 python tools/fuzz_gpu_mutated_db.py $seed 45 > /tmp/fuzz.log 2>&1; echo "seed $seed rc $?" | tee -a $out/fuzz_databases.txt
  grep -v "registering" /tmp/fuzz.log | tail -25 | tee -a $out/fuzz_databases.txt; tail -2 /tmp/fuzz.log | grep registering | tee -a $out/fuzz_databases.txt
done
timeout 1200 python -m pytest tests -m gpu -x -q -n 2 2>&1 | tail -6 | tee $out/gpu_suite.txt
