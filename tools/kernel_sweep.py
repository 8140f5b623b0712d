#!/usr/bin/env python3
"""Measurement aid: decode-kernel time for 64k instances of ONE synthetic clip across clip shapes (run on the GPU box)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from acl_amd import runtime, synth  # noqa: E402

CASES = {
    "default (1% raw)": dict(seed=2),
    "no raw": dict(seed=2, raw_fraction=0.0),
    "no raw, no width0": dict(seed=2, raw_fraction=0.0, width0_fraction=0.0),
    "50% animated rotations": dict(seed=2, rotation_constant=0.48),
    "50% animated, no raw": dict(seed=2, rotation_constant=0.48, raw_fraction=0.0),
    "stripped keyframes": dict(seed=2, strip_keyframes=1),
    "wrap": dict(seed=2, wrap=1),
    "600 samples": dict(seed=2, num_samples=600),
    "31 samples (one segment)": dict(seed=2, num_samples=31),
    "70 bones": dict(seed=2, num_tracks=70),
    "106 bones": dict(seed=2, num_tracks=106),
    "cinematic 300 bones, 451 samples": dict(seed=4, num_tracks=300, num_samples=451, has_scale=1, scale_default=0.7, scale_constant=0.1, rotation_constant=0.45, translation_constant=0.8),
    "cinematic, one segment": dict(seed=4, num_tracks=300, num_samples=31, has_scale=1, scale_default=0.7, scale_constant=0.1, rotation_constant=0.45, translation_constant=0.8),
    "cinematic, 4 segments": dict(seed=4, num_tracks=300, num_samples=64, has_scale=1, scale_default=0.7, scale_constant=0.1, rotation_constant=0.45, translation_constant=0.8),
    "300 bones CMU mix": dict(seed=4, num_tracks=300, num_samples=301),
    "300 bones CMU mix, one segment": dict(seed=4, num_tracks=300, num_samples=31),
    "200 bones CMU mix": dict(seed=4, num_tracks=200, num_samples=301),
    "300 bones all constant": dict(seed=4, num_tracks=300, num_samples=31, rotation_constant=0.98, translation_constant=0.98),
}


def main():
    only = sys.argv[1:] 
    device = torch.device("cuda:0")
    n = 65536
    rng = np.random.default_rng(0)
    for name, spec in CASES.items():
        if only and not any(key in name for key in only):
            continue
        ctx = runtime.Context(0)
        clip = synth.build_clip(**spec)
        handle = ctx.register_clip(clip.blob)
        info = ctx.clip_info(handle)
        times = torch.from_numpy(rng.uniform(0, info.duration, size=n).astype(np.float32)).to(device)
        ids = torch.full((n,), handle, dtype=torch.int32, device=device)
        poses = torch.empty((n, info.num_tracks * 12), dtype=torch.float32, device=device)
        stream = torch.cuda.current_stream(device).cuda_stream
        ms = min(ctx.time_decompress_tracks_batch(ids.data_ptr(), times.data_ptr(), n, poses.data_ptr(), info.num_tracks * 48, repeats=50, stream=stream) for _ in range(3))
        gbps = n * info.num_tracks * 48 / (ms * 1e-3) / 1e9
        print(f"{name:32s} tracks={info.num_tracks:4d} animated={info.num_animated_sub_tracks:4d} segments={info.num_segments:3d}  {ms * 1000:7.1f} us  {gbps:7.0f} GB/s")
        ctx.close()


if __name__ == "__main__":
    main()
