cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 600 python -m pytest tests/test_gpu_order_device.py tests/test_gpu_lifetime.py -x -q 2>&1 | tail -5
python - <<PY
import bench, json
for name in ("256_clips", "one_clip"):
    r = bench.measure_job(name, 0, 0, order="device", repeats=100)
    print(name, {k: r[k] for k in ("kernel_ms", "ordering_ms_device", "frac")})
PY
