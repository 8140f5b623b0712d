# one launch ordering: correctness, time
timeout 900 python -m pytest tests/test_gpu_order_device.py tests/test_gpu_instance_lists.py -m gpu -x -q 2>&1 | tail -5
python tools/order_time.py 2>&1 | tail -1
ACLHIP_ORDER_LAUNCHES=3 python tools/order_time.py 2>&1 | tail -1
timeout 300 python bench.py --workload 256_clips --order list --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | cut -c1-400
