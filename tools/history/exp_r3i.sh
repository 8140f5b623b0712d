# the database workload in list / locality order
run() {
  timeout 300 python bench.py --steps 600 --warmup 100 --no-cpu-baseline --no-extras "$@" 2>&1 | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', round(d['roofline']['kernel_ms']*1000,1), round(d['roofline']['frac'],3))"
}
run --workload database
run --workload database --order locality
run --workload database --order list
run --workload cinematic --order list
