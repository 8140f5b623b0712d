#!/bin/bash
# round 6, one box: the mutated-database fuzz after the one-chunk-per-segment refusal, the GPU suite, then every workload's evidence (tools/profile_round6.sh)
out=gpurun_out/r06g
mkdir -p $out/fuzz
for seed in 72 75 76 77; do
  FUZZ_VERBOSE=1 FUZZ_SAVE_DIR=$out/fuzz timeout 200 python tools/fuzz_gpu_mutated_db.py $seed 60 > /tmp/fuzz.log 2>&1; echo "seed $seed rc $?" | tee -a $out/fuzz_databases.txt
  grep -v "registering" /tmp/fuzz.log | tail -12 | tee -a $out/fuzz_databases.txt
done
timeout 1200 python -m pytest tests -m gpu -x -q -n 2 2>&1 | tail -4 | tee $out/gpu_suite.txt
bash tools/profile_round6.sh r06 2>&1 | tail -60 | tee $out/profile_tail.txt
