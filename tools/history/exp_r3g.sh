# persistent instance list, 1 % of the instances change clip per step: when to re-order (ACLHIP_LIST_REORDER_DIVISOR: once 1/divisor has changed)
run() {
  timeout 300 python bench.py --steps 1000 --warmup 200 --no-cpu-baseline --no-extras "$@" 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$LABEL', '$*', round(d['ms_per_step']*1000,1), round(d['roofline']['kernel_ms']*1000,1), round(d['roofline']['frac'],3))"
}
for divisor in 8 16 32 64 128; do
  export ACLHIP_LIST_REORDER_DIVISOR=$divisor LABEL="divisor=$divisor"
  run --workload 256_clips --order list
done
