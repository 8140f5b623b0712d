#!/usr/bin/env python3
"""aclhip_order_instances_device on the 256_clips batch: mean time of back-to-back calls and of the step (order + decode).
ACLHIP_ORDER_LAUNCHES / ACLHIP_ORDER_GRID_LOG2_BLOCKS pick the form (read once per process)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

entry = bench.measure_job("256_clips", 0, 0, repeats=400, order="device")
print(json.dumps({"form": os.environ.get("ACLHIP_ORDER_LAUNCHES", "default"), "log2_blocks": os.environ.get("ACLHIP_ORDER_GRID_LOG2_BLOCKS"),
                  "ordering_us": round(entry["ordering_ms_device"] * 1000, 2), "step_us": round(entry["kernel_ms"] * 1000, 2),
                  "decode_us_order_reused": round(entry["kernel_ms_order_reused"] * 1000, 2)}))
