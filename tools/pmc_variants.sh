#!/bin/bash
# Measurement aid (round 3): SQ / TA counters of the 300-bone rig workload under each experimental kernel variant -> gpurun_out/pmc_variants.txt
# usage: tools/pmc_variants.sh   (library built with -DACLHIP_EXPERIMENTS; ACLHIP_LIBRARY may point at it)
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export ACLHIP_LIBRARY=${ACLHIP_LIBRARY:-acl_amd/lib/libaclhip_exp.so}   # tools/build_experiments.sh
mkdir -p gpurun_out
out=gpurun_out/pmc_variants.txt
: > $out
GROUPS_LIST=(
  "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_INST_ANY"
  "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_WAIT_ANY"
  "GRBM_GUI_ACTIVE TA_TA_BUSY_sum TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum TCC_EA0_WRREQ_STALL_sum"
)
run_variant() {
  label=$1; shift
  echo "== $label ($*)" >> $out
  g=0
  for group in "${GROUPS_LIST[@]}"; do
    dir=/tmp/pmc_var_$g
    rm -rf $dir
    env "$@" ACLHIP_BENCH_PROFILING=1 timeout 150 rocprofv3 --pmc $group --output-format csv -d $dir -o pass -- python bench.py --workload ${WORKLOAD:-cinematic} --steps 10 --warmup 2 --no-cpu-baseline --no-extras > /tmp/pmc_log_$g.txt 2>&1
    csv=$(find $dir -name "*counter_collection.csv" | head -1)
    if [ -n "$csv" ]; then python tools/pmc_summary.py decompress $csv | sed "s#^.*csv: ##" >> $out; else echo "group $g failed" >> $out; fi
    g=$((g+1))
  done
  env "$@" python bench.py --workload ${WORKLOAD:-cinematic} --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('kernel_ms', d['roofline']['kernel_ms'], 'frac', round(d['roofline']['frac'],4))" >> $out
}
run_variant "shipped: one aligned 16 byte read per key" ACLHIP_WIDE_KEY_LOADS=1
run_variant "round 2 kernel: 8 + 4 byte reads at byte addresses" ACLHIP_WIDE_KEY_LOADS=0
run_variant "decode waves + one store wave per workgroup (7 + 1), one-shot" ACLHIP_HANDOFF_DECODERS=7
run_variant "last arriver of 4 stores the workgroup's windows" ACLHIP_HANDOFF_DECODERS=4 ACLHIP_HANDOFF_LAST_ARRIVER=1
run_variant "persistent: 7 decode waves + 1 store wave x 4 per CU" ACLHIP_PERSISTENT=71
run_variant "4 work items per wave in turn" ACLHIP_ITEMS_PER_WAVE=4
run_variant "staged: keyframe runs through LDS, one base pose image per workgroup of 4" ACLHIP_STAGED=4
cat $out
