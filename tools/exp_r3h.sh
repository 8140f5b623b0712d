# scalar lists: what bounds the float1f kernel -- timing only variants (wrong results): no stores / no field arithmetic.
# The two libraries were built from the hand-written float1f decode of round 3 with -DACLHIP_EXP_SCALAR_NO_STORES / _NO_DECODE; that
# decode is not kept (same time as the shipped one), so this script documents the run rather than reproducing it.
run() {
timeout 300 python bench.py --workload scalar --steps 600 --warmup 100 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['roofline']['kernel_ms']*1000,2), round(d['roofline']['frac'],3))"
}
run shipped
ACLHIP_LIBRARY=$PWD/acl_amd/lib/libaclhip_ns.so run "no stores"
ACLHIP_LIBRARY=$PWD/acl_amd/lib/libaclhip_nd.so run "no decode arithmetic"
