#!/usr/bin/env python3
"""Measurement aid: decompress_track (single bone requests) throughput on the GPU box."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from acl_amd import runtime, synth  # noqa: E402


def main():
    ctx = runtime.Context(0)
    clip = synth.build_clip(seed=2, num_tracks=100, num_samples=301, sample_rate=30.0)
    handle = ctx.register_clip(clip.blob)
    rng = np.random.default_rng(0)
    for n in (65536, 1 << 20, 1 << 22):
        times = torch.from_numpy(rng.uniform(0, clip.duration, size=n).astype(np.float32)).cuda()
        tracks = torch.from_numpy(rng.integers(0, 100, size=n).astype(np.int32)).cuda()
        ids = torch.full((n,), handle, dtype=torch.int32, device="cuda")
        out = torch.empty((n, 12), dtype=torch.float32, device="cuda")
        stream = torch.cuda.current_stream()
        for _ in range(50):
            ctx.decompress_track_batch(ids.data_ptr(), times.data_ptr(), tracks.data_ptr(), n, out.data_ptr(), stream=stream.cuda_stream)
        start, stop = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        start.record(stream)
        for _ in range(100):
            ctx.decompress_track_batch(ids.data_ptr(), times.data_ptr(), tracks.data_ptr(), n, out.data_ptr(), stream=stream.cuda_stream)
        stop.record(stream)
        stop.synchronize()
        us = start.elapsed_time(stop) / 100 * 1000
        print(f"decompress_track: {n:8d} random (instance, bone) requests  {us:8.1f} us  {n / us / 1e3:7.2f} G bones/s  {n * 48 / us / 1e3:7.0f} GB/s written")
    ctx.close()


if __name__ == "__main__":
    main()
