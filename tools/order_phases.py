#!/usr/bin/env python3
"""Where a launch of order_instances_grid_kernel spends its time: wall clock stamps per workgroup phase (experiments build,
ACLHIP_LIBRARY=acl_amd/lib/libaclhip_exp.so). Prints, over the workgroups of the last of a few launches, mean / max microseconds
since the earliest workgroup start."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import numpy as np

job = bench.Job("256_clips", 0, 0, order="device")
job.prewarm(0.02)
for _ in range(20):
    job.order_step()
job.torch.cuda.synchronize()
lib = ctypes.CDLL(os.environ["ACLHIP_LIBRARY"])
stamps = np.zeros((64, 8), dtype=np.uint64)
assert lib.aclhip_exp_read_order_stamps(stamps.ctypes.data_as(ctypes.c_void_p)) == 0
used = stamps[stamps[:, 0] != 0].astype(np.int64)
origin = used[:, 0].min()
names = ["start", "column written", "share of rows scanned", "first barrier passed", "second barrier passed", "cursors", "placed", "stores acknowledged"]
print("workgroups", len(used), "log2_blocks", os.environ.get("ACLHIP_ORDER_GRID_LOG2_BLOCKS"))
for k, name in enumerate(names):
    t = (used[:, k] - origin) * 0.01
    print(f"  {name:22s} mean {t.mean():6.2f} us   max {t.max():6.2f} us")
job.close()
