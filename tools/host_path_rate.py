#!/usr/bin/env python3
"""Measurement aid: the PCIe-inclusive rate of the host pointer convenience entry point (aclhip_decompress_tracks_host: stage the
instance list, launch, copy the poses back) for the headline batch, next to the device-resident rate bench.py reports."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from acl_amd import runtime, synth  # noqa: E402

clip = synth.build_clip(seed=2, num_tracks=100, num_samples=301, sample_rate=30.0)
with runtime.Context(0) as context:
    handle = context.register_clip(clip.blob)
    n = 65536
    rng = np.random.default_rng(0)
    times = rng.uniform(0.0, clip.duration, size=n).astype(np.float32)
    handles = np.full(n, handle, dtype=np.uint32)
    out = np.zeros((n, 100, 12), dtype=np.float32)
    context.decompress_tracks(handles, times, out=out, num_tracks=100)          # first call: allocations, page faults
    runs = []
    for _ in range(5):
        t0 = time.perf_counter()
        context.decompress_tracks(handles, times, out=out, num_tracks=100)
        runs.append(time.perf_counter() - t0)
    best = min(runs)
    print(f"aclhip_decompress_tracks_host, {n} instances x 100 bones, pageable host memory: {best * 1e3:.1f} ms per call "
          f"= {n / best / 1e6:.1f} M poses/s, {n * 4800 * 2 / best / 1e9:.1f} GB/s over PCIe (poses up and down)")
