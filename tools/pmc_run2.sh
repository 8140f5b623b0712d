#!/bin/bash
# Measurement aid (round 3): rocprofv3 PMC passes (counters only, one group per run) over bench.py workloads -> gpurun_out/pmc_<tag>_<workload>.txt
# usage: tools/pmc_run2.sh <tag> "<workload> [<workload>...]" [extra bench args]
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
tag=$1
GROUPS_LIST=(
  "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_INST_ANY"
  "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAIT_ANY"
  "SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQ_INST_CYCLES_SMEM SQ_ACTIVE_INST_LDS SQ_INST_LEVEL_VMEM"
  "GRBM_GUI_ACTIVE SQ_LEVEL_WAVES SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_BUSY_CU_CYCLES SQ_INST_LEVEL_SMEM SQ_INST_LEVEL_LDS"
  "TCC_HIT_sum TCC_MISS_sum TCC_READ_sum TCC_WRITE_sum"
  "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_STALL_sum TCC_EA0_RDREQ_sum"
  "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"
  "TA_BUSY_avr TA_TA_BUSY_sum TCP_GATE_EN1_sum TCP_TA_TCP_STATE_READ_sum"
  "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_LATENCY_sum"
  "SQ_WAIT_INST_LDS SQ_INSTS_FLAT SQ_THREAD_CYCLES_VALU SQ_IFETCH SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_MISC"
)
for workload in $2; do
  out=gpurun_out/pmc_${tag}_${workload}.txt
  : > $out
  g=0
  for group in "${GROUPS_LIST[@]}"; do
    dir=/tmp/pmc_${workload}_$g
    rm -rf $dir
    ACLHIP_BENCH_PROFILING=1 timeout 150 rocprofv3 --pmc $group --output-format csv -d $dir -o pass -- python bench.py --workload $workload --steps 10 --warmup 2 --no-cpu-baseline --no-extras ${3:-} > /tmp/pmc_log_$g.txt 2>&1
    csv=$(find $dir -name "*counter_collection.csv" | head -1)
    if [ -n "$csv" ]; then python tools/pmc_summary.py decompress $csv | sed "s#^.*csv: ##" >> $out; else echo "group $g ($group) failed: $(grep -i "error\|invalid\|not" /tmp/pmc_log_$g.txt | head -2)" >> $out; fi
    g=$((g+1))
  done
  echo "== $workload"; cat $out
done
