#!/usr/bin/env python3
"""Measurement aid: scalar track list kernel time for 64k instances across list shapes (run on the GPU box)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from acl_amd import runtime, synth  # noqa: E402

CASES = {
    "float1f x256, 10% raw": dict(seed=9, track_type=0, num_tracks=256, num_samples=120, sample_rate=60.0),
    "float1f x256, 1% raw": dict(seed=9, track_type=0, num_tracks=256, num_samples=120, sample_rate=60.0, raw_fraction=0.01),
    "float1f x256, no raw": dict(seed=9, track_type=0, num_tracks=256, num_samples=120, sample_rate=60.0, raw_fraction=0.0),
    "float1f x64, no raw": dict(seed=9, track_type=0, num_tracks=64, num_samples=120, sample_rate=60.0, raw_fraction=0.0),
    "float3f x256, no raw": dict(seed=9, track_type=2, num_tracks=256, num_samples=120, sample_rate=60.0, raw_fraction=0.0),
    "vector4f x256, no raw": dict(seed=9, track_type=4, num_tracks=256, num_samples=120, sample_rate=60.0, raw_fraction=0.0),
    "float1f x1024, no raw": dict(seed=9, track_type=0, num_tracks=1024, num_samples=60, sample_rate=60.0, raw_fraction=0.0),
}


def main():
    n = 65536
    rng = np.random.default_rng(0)
    for name, spec in CASES.items():
        ctx = runtime.Context(0)
        clip = synth.build_scalar_clip(**spec)
        handle = ctx.register_clip(clip.blob)
        row = clip.num_tracks * clip.num_components
        times = torch.from_numpy(rng.uniform(0, clip.duration, size=n).astype(np.float32)).cuda()
        ids = torch.full((n,), handle, dtype=torch.int32, device="cuda")
        out = torch.empty((n, row), dtype=torch.float32, device="cuda")
        stream = torch.cuda.current_stream()
        for _ in range(200):
            ctx.decompress_scalar_tracks_batch(ids.data_ptr(), times.data_ptr(), n, out.data_ptr(), row * 4, stream=stream.cuda_stream)
        start, stop = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        start.record(stream)
        for _ in range(500):
            ctx.decompress_scalar_tracks_batch(ids.data_ptr(), times.data_ptr(), n, out.data_ptr(), row * 4, stream=stream.cuda_stream)
        stop.record(stream)
        stop.synchronize()
        us = start.elapsed_time(stop) / 500 * 1000
        print(f"{name:28s} {us:7.1f} us  {n * row * 4 / us / 1e3:7.0f} GB/s written  {n * row / us / 1e3:6.1f} G values/s")
        ctx.close()


if __name__ == "__main__":
    main()
