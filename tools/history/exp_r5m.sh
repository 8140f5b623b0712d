#!/bin/bash
# round 5, set M: clips mutated until the validators accept them, registered and decoded on the GPU against the oracle (tools/fuzz_gpu_mutated.py)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r5m
for seed in 3; do
  FUZZ_SAVE_DIR=$PWD/gpurun_out/r5m timeout 100 python tools/fuzz_gpu_mutated.py $seed 6 > gpurun_out/r5m/seed_$seed.log 2>&1
  echo "seed $seed rc=$? $(grep -E 'gpu mutated fuzz' gpurun_out/r5m/seed_$seed.log)"
done | tee gpurun_out/r5m/summary.txt
