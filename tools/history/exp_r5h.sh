#!/bin/bash
# round 5, set H (one gpurun call): scalar track lists taken in turn -- parity, then groups of 1 / 2 / 4 instances x 1 / 2 / 4 / 8 turns per wave
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_scalar.py tests/test_gpu_instance_writers.py -m gpu -x -q > gpurun_out/r5h_pytest.txt 2>&1; tail -3 gpurun_out/r5h_pytest.txt
for g in lab g2 g1; do
  L=ACLHIP_LIBRARY=$PWD/acl_amd/lib/libaclhip_$g.so
  ACLHIP_LIBRARY=$PWD/acl_amd/lib/libaclhip_$g.so timeout 300 python -m pytest tests/test_gpu_scalar.py -m gpu -x -q 2>&1 | tail -1
  python tools/variant_sweep.py --repeats 400 --workloads scalar $g-auto:$L $g-t1:$L,ACLHIP_SCALAR_TURNS=1 $g-t2:$L,ACLHIP_SCALAR_TURNS=2 $g-t4:$L,ACLHIP_SCALAR_TURNS=4 $g-t8:$L,ACLHIP_SCALAR_TURNS=8 $g-t16:$L,ACLHIP_SCALAR_TURNS=16 | tee -a gpurun_out/r5h_scalar_turns.txt
done
python bench.py --workload database --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('database', d['roofline']['kernel_ms'], d.get('self_check'))"
python bench.py --workload one_clip --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('one_clip', d['roofline']['kernel_ms'], d.get('self_check'))"
python bench.py --workload one_clip --layout qv32 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('one_clip qv32', d['roofline']['kernel_ms'], d.get('self_check'))"
python bench.py --workload track_requests --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('track_requests', d['roofline']['kernel_ms'], d.get('self_check'))"
python bench.py --workload scalar --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('scalar', d['roofline']['kernel_ms'], d.get('self_check'))"
python - <<'PY'
import bench, json
print(json.dumps(bench.measure_database_paging(0, 0)))
PY
