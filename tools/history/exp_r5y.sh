#!/bin/bash
# round 5, set Y: what a pose kernel compiled for a PLAIN launch (one window per pose, no instance list, no per instance arrays, no
# stripped key frames, no database -- -DACLHIP_EXP_PLAIN, an experiment build: only such launches decode correctly with it) and a base
# pose DMA without per row lane masks would buy: time, and instructions per wave
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r5y
L=$PWD/acl_amd/lib
V="shipped plain:ACLHIP_LIBRARY=$L/libaclhip_plain.so shipped2 plain2:ACLHIP_LIBRARY=$L/libaclhip_plain.so shipped3 plain3:ACLHIP_LIBRARY=$L/libaclhip_plain.so"
python tools/variant_sweep.py --repeats 300 --workloads one_clip,256_clips $V | tee gpurun_out/r5y/sweep.txt
python tools/variant_sweep.py --repeats 300 --workloads one_clip --layout qv32 $V | tee -a gpurun_out/r5y/sweep.txt
python tools/variant_sweep.py --repeats 300 --workloads one_clip --layout qvv40 $V | tee -a gpurun_out/r5y/sweep.txt
python tools/variant_sweep.py --repeats 300 --workloads 256_clips --order locality $V | tee -a gpurun_out/r5y/sweep.txt
for lib in "" $L/libaclhip_plain.so; do
  rm -rf /tmp/prof_sq
  ACLHIP_LIBRARY=$lib ACLHIP_BENCH_PROFILING=1 timeout 200 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM --output-format csv -d /tmp/prof_sq -o pass -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras > /dev/null 2> /tmp/sq.log
  csv=$(find /tmp/prof_sq -name "*counter_collection.csv" | head -1)
  echo "== ${lib:-shipped}" | tee -a gpurun_out/r5y/insts.txt
  [ -n "$csv" ] && python tools/pmc_summary.py decompress $csv | sed "s#^.*csv: ##" | tee -a gpurun_out/r5y/insts.txt
done
