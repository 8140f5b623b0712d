#!/bin/bash
# round 5, experiment set B (one gpurun call): the whole GPU suite (ordering give-up path, lab library), then decompress_track_kernel with
# 5 KiB of LDS per wave at 7 waves per SIMD (shipped) and 8 (libaclhip_w8.so: 15 spilled registers), then its SQ / TA counters
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -5
echo "== 7 waves per SIMD"; TRACK_SWEEP_SIZES=4194304 timeout 300 python tools/track_sweep.py 2>&1 | tee gpurun_out/r5b_track_sweep_w7.txt
echo "== 8 waves per SIMD"; TRACK_SWEEP_SIZES=4194304 ACLHIP_LIBRARY=$PWD/acl_amd/lib/libaclhip_w8.so timeout 300 python tools/track_sweep.py 2>&1 | tee gpurun_out/r5b_track_sweep_w8.txt
python bench.py --workload track_requests --no-cpu-baseline --no-extras 2>&1 | tail -1 | tee gpurun_out/r5b_track_bench.json
ACLHIP_LIBRARY=$PWD/acl_amd/lib/libaclhip_w8.so python bench.py --workload track_requests --no-cpu-baseline --no-extras 2>&1 | tail -1 | tee gpurun_out/r5b_track_bench_w8.json
sq=gpurun_out/r5b_track_requests_pmc_sq.txt
: > $sq
for group in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_INST_ANY" \
             "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_WAIT_ANY" \
             "GRBM_GUI_ACTIVE TA_TA_BUSY_sum TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum TCC_EA0_WRREQ_STALL_sum" \
             "FETCH_SIZE" "WRITE_SIZE"; do
  rm -rf /tmp/prof_sq
  ACLHIP_BENCH_PROFILING=1 timeout 200 rocprofv3 --pmc $group --output-format csv -d /tmp/prof_sq -o pass -- python bench.py --workload track_requests --steps 10 --warmup 2 --no-cpu-baseline --no-extras > /dev/null 2> /tmp/sq.log
  csv=$(find /tmp/prof_sq -name "*counter_collection.csv" | head -1)
  [ -n "$csv" ] && python tools/pmc_summary.py decompress $csv | sed "s#^.*csv: ##" >> $sq
done
cat $sq
