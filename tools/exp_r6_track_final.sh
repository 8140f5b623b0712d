#!/bin/bash
# round 6: the shipped track kernel (record heads gathered four lanes per record) -- parity, every request pattern, the counters of the mixed-clip pattern before / after, and its evidence files again. Output: gpurun_out/r06i/
out=gpurun_out/r06i
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 1200 python -m pytest tests -m gpu -x -q -n 2 2>&1 | tail -4 | tee $out/gpu_suite.txt
export TRACK_SWEEP_SIZES=4194304
timeout 300 python tools/track_sweep.py 2>&1 | grep decompress_track | tee $out/track_patterns.txt
# counters of the mixed-clip pattern: one pattern per process (TRACK_SWEEP_ONLY), previous build against the shipped one
for v in ab_prev current; do
  lib=$PWD/acl_amd/lib/libaclhip_$v.so; [ $v = current ] && lib=$PWD/acl_amd/lib/libaclhip.so
  for group in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" "GRBM_GUI_ACTIVE TA_TA_BUSY_sum TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum" "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
    rm -rf /tmp/prof_mixed
    ACLHIP_LIBRARY=$lib TRACK_SWEEP_ONLY="256 clips as drawn, random bones" TRACK_SWEEP_REPEATS=10 timeout 300 rocprofv3 --pmc $group --output-format csv -d /tmp/prof_mixed -o pass -- python tools/track_sweep.py > /dev/null 2> /tmp/mixed.log
    csv=$(find /tmp/prof_mixed -name "*counter_collection.csv" | head -1)
    [ -n "$csv" ] && python tools/pmc_summary.py decompress_track_kernel $csv | sed "s#^.*csv: ##" | sed "s/^/$v: /" | tee -a $out/mixed_pattern_pmc.txt
  done
done
bash tools/profile_round6.sh r06 track_requests 2>&1 | tail -5
