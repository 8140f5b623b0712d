#!/bin/bash
# round 6: the mutated-clip fuzz with the full-format sources, the mutated-database fuzz (verbose: a crash names its mutation), the GPU suite. Output: gpurun_out/r06e/
out=gpurun_out/r06e
mkdir -p $out/fuzz
for seed in 61 62 63; do
  FUZZ_VERBOSE=1 FUZZ_SAVE_DIR=$out/fuzz timeout 200 python tools/fuzz_gpu_mutated.py $seed 45 > /tmp/fuzz.log 2>&1; echo "seed $seed rc $?" | tee -a $out/fuzz_clips.txt
  grep -v "registering" /tmp/fuzz.log | tail -5 | tee -a $out/fuzz_clips.txt; tail -2 /tmp/fuzz.log | grep registering | tee -a $out/fuzz_clips.txt
done
FUZZ_SCALAR=1 FUZZ_VERBOSE=1 timeout 200 python tools/fuzz_gpu_mutated.py 64 30 > /tmp/fuzz.log 2>&1; echo "scalar seed 64 rc $?" | tee -a $out/fuzz_clips.txt
grep -v "registering" /tmp/fuzz.log | tail -3 | tee -a $out/fuzz_clips.txt
for seed in 71 72 73; do
  FUZZ_VERBOSE=1 FUZZ_SAVE_DIR=$out/fuzz timeout 200 python tools/fuzz_gpu_mutated_db.py $seed 45 > /tmp/fuzz.log 2>&1; echo "seed $seed rc $?" | tee -a $out/fuzz_databases.txt
  grep -v "registering" /tmp/fuzz.log | tail -5 | tee -a $out/fuzz_databases.txt; tail -2 /tmp/fuzz.log | grep registering | tee -a $out/fuzz_databases.txt
done
timeout 1200 python -m pytest tests -m gpu -x -q -n 2 2>&1 | tail -6 | tee $out/gpu_suite.txt
