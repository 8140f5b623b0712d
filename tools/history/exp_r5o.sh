#!/bin/bash
# round 5, set O: aclhip_order_instances_device on two streams + a captured graph while three other processes keep the device busy
# (tools/order_under_load.py): the graph replayed on its capture stream, and -- the misuse the scatter guards are for -- on the default
# stream next to a plain call that uses the same scratch; then the ordering tests
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r5s2
for i in 1 2 3; do (timeout 150 python bench.py --workload cinematic --steps 500000 --warmup 10 --no-extras --no-cpu-baseline > /dev/null 2>&1 &) ; done
sleep 20
for v in "X=0" "X=0" "ORDER_REPLAY_ON_CURRENT_STREAM=1" "ORDER_REPLAY_ON_CURRENT_STREAM=1" "ORDER_REPLAY_ON_CURRENT_STREAM=1"; do
  echo "== $v"; env $v timeout 120 python tools/order_under_load.py 4 8 2>&1 | grep -v amdgpu.ids | grep -E "replay [0-9]+:|fault|done|NOT|identity" | grep -v launching | tail -n 12
done 2>&1 | tee gpurun_out/r5s2/order_under_load.txt
timeout 200 python -m pytest tests/test_gpu_order_device.py tests/test_gpu_instance_lists.py -x -q 2>&1 | grep -E "passed|failed|Aborted" | tee -a gpurun_out/r5s2/order_under_load.txt
