# scalar lists: what the 17 us that are neither arithmetic nor stores are made of -- timing only variants (wrong results).
# Built from temporary -DACLHIP_EXP_SCALAR_NO_TABLES / _NO_DMA switches in kernels_scalar.inl that are not kept: this script
# documents the run rather than reproducing it.
run() {
timeout 300 python bench.py --workload scalar --steps 600 --warmup 100 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['roofline']['kernel_ms']*1000,2))"
}
run shipped
ACLHIP_LIBRARY=$PWD/acl_amd/lib/libaclhip_nt.so run "no table loads"
ACLHIP_LIBRARY=$PWD/acl_amd/lib/libaclhip_nm.so run "no frame copies into LDS"
ACLHIP_LIBRARY=$PWD/acl_amd/lib/libaclhip_ntm.so run "neither"
