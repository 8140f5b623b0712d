#!/usr/bin/env python3
"""Measurement aid (library built with -DACLHIP_EXP_PHASE_TIMES, ACLHIP_LIBRARY pointing at it, ACLHIP_PERSISTENT=<shape>): where a decode wave
of the persistent pose kernel spends its time per work item (100 MHz wall clock). usage: persistent_times.py <workload>"""
import ctypes
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench


def main():
    workload = sys.argv[1] if len(sys.argv) > 1 else "cinematic"
    job = bench.Job(workload, 0, 0)
    job.prewarm(0.05)
    for _ in range(20):
        job.step()
    job.torch.cuda.synchronize()
    stamps = np.zeros(16384 * 4, dtype=np.uint64)
    status = job.lib.aclhip_debug_read_phase_times(ctypes.c_void_p(stamps.ctypes.data), ctypes.c_uint32(stamps.size))
    assert status == 0, status
    t = stamps.reshape(-1, 8).astype(np.int64)
    t = t[t[:, 3] != 0]
    items = t[:, 3]
    print(workload, "decoders", t.shape[0], "items per decoder %.1f .. %.1f" % (items.min(), items.max()))
    for name, column in (("scalar prologue + seek", 0), ("tables + keyframes + unpack + wait", 1), ("publish + next image", 2)):
        per_item = t[:, column] / items * 0.01
        print("  %-36s mean %6.2f  p10 %6.2f  p50 %6.2f  p90 %6.2f us per item" % (name, per_item.mean(), np.percentile(per_item, 10), np.percentile(per_item, 50), np.percentile(per_item, 90)))
    life = (t[:, 5] - t[:, 4]) * 0.01
    print("  decoder life mean %.1f  min %.1f  max %.1f us; first start .. last end %.1f us" % (life.mean(), life.min(), life.max(), (t[:, 5].max() - t[:, 4].min()) * 0.01))


if __name__ == "__main__":
    main()
