#!/bin/bash
# round 6: the fuzzers on the final build (random shapes x settings x consumers, exact math analysis, mutated clips / scalar lists / databases). Output: gpurun_out/r06o/
out=gpurun_out/r06o
mkdir -p $out/fuzz
for seed in 101 102; do
  timeout 300 python tools/fuzz_gpu.py 150 $seed 2>&1 | tail -2 | tee -a $out/fuzz_random.txt
done
timeout 200 python tools/fuzz_exact_math.py 60 103 2>&1 | tail -1 | tee -a $out/fuzz_random.txt
for seed in 111 112; do
  FUZZ_SAVE_DIR=$out/fuzz timeout 300 python tools/fuzz_gpu_mutated.py $seed 120 2>&1 | tail -2 | tee -a $out/fuzz_clips.txt
done
FUZZ_SCALAR=1 timeout 200 python tools/fuzz_gpu_mutated.py 113 60 2>&1 | tail -1 | tee -a $out/fuzz_clips.txt
for seed in 121 122; do
  FUZZ_SAVE_DIR=$out/fuzz timeout 300 python tools/fuzz_gpu_mutated_db.py $seed 120 2>&1 | tail -2 | tee -a $out/fuzz_databases.txt
done
