#!/usr/bin/env python3
"""Measurement aid (needs a library built with -DACLHIP_EXP_PHASE_TIMES, ACLHIP_LIBRARY pointing at it): wall clock stamps, 100 MHz clock, of
the pose consumer kernel's phases per workgroup (entry, poses decoded, walk done, stores issued) or of the pose kernel's phases for the
first wave of the first 16384 workgroups (entry, seek done, window decoded into LDS, stores issued). usage: phase_times.py <workload> [layout]"""
import ctypes
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench


def main():
    workload = sys.argv[1] if len(sys.argv) > 1 else "object_space"
    layout = sys.argv[2] if len(sys.argv) > 2 else "qvv48"
    job = bench.Job(workload, 0, 0, layout=layout)
    job.prewarm(0.05)
    for _ in range(20):
        job.step()
    job.torch.cuda.synchronize()
    blocks = 16384
    stamps = np.zeros(blocks * 4, dtype=np.uint64)
    status = job.lib.aclhip_debug_read_phase_times(ctypes.c_void_p(stamps.ctypes.data), ctypes.c_uint32(stamps.size))
    assert status == 0, status
    stamps = stamps.reshape(blocks, 4).astype(np.int64)
    stamps = stamps[stamps[:, 0] != 0]
    t = (stamps - stamps[:, 0].min()) * 0.01        # us
    print(workload, layout, "workgroups", t.shape[0], "kernel span %.1f us" % (t[:, 3].max()))
    consumer = workload in ("object_space", "additive_object_space")
    names = ("decode", "walk", "store issue") if consumer else ("seek", "tables + keyframes + unpack", "store issue")
    for name, a, b in ((names[0], 0, 1), (names[1], 1, 2), (names[2], 2, 3), ("life", 0, 3)):
        d = t[:, b] - t[:, a]
        print("  %-12s mean %6.2f  p10 %6.2f  p50 %6.2f  p90 %6.2f us" % (name, d.mean(), np.percentile(d, 10), np.percentile(d, 50), np.percentile(d, 90)))
    # concurrency: workgroups alive at sample points
    points = np.linspace(0, t[:, 3].max(), 50)[5:-5]
    alive = [(np.sum((t[:, 0] <= p) & (t[:, 3] > p))) for p in points]
    print("  stamped workgroups in flight (whole GPU): mean %.0f = %.2f per CU" % (np.mean(alive), np.mean(alive) / 256))
    for name, a, b in ((names[0], 0, 1), (names[1], 1, 2), (names[2], 2, 3)):
        n = [(np.sum((t[:, a] <= p) & (t[:, b] > p))) for p in points]
        print("    %-9s %.2f per CU" % (name, np.mean(n) / 256))


if __name__ == "__main__":
    main()
