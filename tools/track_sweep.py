#!/usr/bin/env python3
"""Measurement aid: decompress_track (single bone requests) throughput on the GPU box, per request pattern:
one clip / 256 clips as drawn / 256 clips in runs of 64 requests per clip / the root bone of every instance of a 256-clip crowd.
Every pattern is first checked against the whole-pose kernel (decompress_track == decompress_tracks, bit for bit)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from acl_amd import runtime, synth  # noqa: E402


FAST = os.environ.get("TRACK_SWEEP_FAST", "0") == "1"       # ACLHIP_DECODE_FAST: rotations within 2e-6, the rest bit identical


def time_requests(ctx, ids, times, tracks, out, repeats=100):
    n = ids.numel()
    stream = torch.cuda.current_stream()
    params = runtime.default_params(flags=runtime.DECODE_FAST if FAST else 0)
    for _ in range(30):
        ctx.decompress_track_batch(ids.data_ptr(), times.data_ptr(), tracks.data_ptr(), n, out.data_ptr(), params=params, stream=stream.cuda_stream)
    start, stop = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    start.record(stream)
    for _ in range(repeats):
        ctx.decompress_track_batch(ids.data_ptr(), times.data_ptr(), tracks.data_ptr(), n, out.data_ptr(), params=params, stream=stream.cuda_stream)
    stop.record(stream)
    stop.synchronize()
    return start.elapsed_time(stop) / repeats * 1000


def check_against_poses(ctx, ids, times, tracks, out, num_tracks, sample=8192):
    """the first `sample` requests against the whole-pose kernel"""
    n = min(sample, ids.numel())
    stream = torch.cuda.current_stream()
    poses = torch.zeros((n, num_tracks, 12), dtype=torch.float32, device="cuda")
    ctx.decompress_tracks_batch(ids.data_ptr(), times.data_ptr(), n, poses.data_ptr(), num_tracks * 48, stream=stream.cuda_stream)
    out.zero_()
    ctx.decompress_track_batch(ids.data_ptr(), times.data_ptr(), tracks.data_ptr(), ids.numel(), out.data_ptr(), params=runtime.default_params(flags=runtime.DECODE_FAST if FAST else 0), stream=stream.cuda_stream)
    stream.synchronize()
    expected = poses[torch.arange(n, device="cuda"), tracks[:n].long()]
    same = torch.equal(expected.view(torch.int32), out[:n].view(torch.int32))
    if FAST:
        lanes = torch.tensor([4, 5, 6, 8, 9, 10], device="cuda")
        same = bool((expected[:, :4] - out[:n, :4]).abs().max() <= 2e-6) and torch.equal(expected[:, lanes].view(torch.int32), out[:n][:, lanes].view(torch.int32))
    nonzero = bool((out.view(torch.int32) != 0).any(dim=1).all())      # every request wrote something (rotations are never all zero)
    return same and nonzero


def main():
    ctx = runtime.Context(0)
    one = synth.build_clip(seed=2, num_tracks=100, num_samples=301, sample_rate=30.0)
    one_handle = ctx.register_clip(one.blob)
    spec_rng = np.random.default_rng(3)
    crowd = []
    for i in range(256):
        animated = spec_rng.uniform(0.25, 0.5)
        crowd.append(synth.build_clip(seed=300 + i, num_tracks=100, num_samples=int(spec_rng.integers(31, 601)), sample_rate=30.0,
                                      rotation_default=0.02, rotation_constant=float(0.98 - animated), wrap=int(spec_rng.uniform() < 0.1),
                                      strip_keyframes=int(spec_rng.uniform() < 0.1), min_bits=int(spec_rng.integers(5, 10)), max_bits=int(spec_rng.integers(12, 19))))
    crowd_handles = np.array([ctx.register_clip(c.blob) for c in crowd], dtype=np.int32)
    crowd_durations = np.array([c.duration for c in crowd], dtype=np.float32)

    rng = np.random.default_rng(0)
    sizes = [int(s) for s in os.environ.get("TRACK_SWEEP_SIZES", "65536,1048576,4194304").split(",")]
    for n in sizes:
        patterns = {}
        patterns["one clip, random bones"] = (np.full(n, one_handle, dtype=np.int32), rng.uniform(0, one.duration, size=n).astype(np.float32), rng.integers(0, 100, size=n).astype(np.int32))
        which = rng.integers(0, 256, size=n)
        patterns["256 clips as drawn, random bones"] = (crowd_handles[which], (rng.uniform(0, 1, size=n).astype(np.float32) * crowd_durations[which]).astype(np.float32), rng.integers(0, 100, size=n).astype(np.int32))
        runs = np.repeat(rng.integers(0, 256, size=(n + 63) // 64), 64)[:n]
        patterns["256 clips in runs of 64, random bones"] = (crowd_handles[runs], (rng.uniform(0, 1, size=n).astype(np.float32) * crowd_durations[runs]).astype(np.float32), rng.integers(0, 100, size=n).astype(np.int32))
        # ... and what a caller who can reorder its requests does: the library's own locality order (the permutation it computes for instance lists)
        permutation = ctx.order_instances_for_locality(crowd_handles[which].astype(np.uint32))
        drawn = patterns["256 clips as drawn, random bones"]
        patterns["256 clips as drawn, requests in aclhip_order_instances_for_locality order"] = tuple(column[permutation] for column in drawn)
        by_clip = np.argsort(which, kind="stable")
        patterns["256 clips as drawn, requests sorted by clip"] = tuple(column[by_clip] for column in drawn)
        # ... sorted by clip AND laid out for the XCDs: workgroup b (256 requests) runs on XCD b % 8, so clip c's requests go to the workgroups
        # of XCD c % 8 -- every clip is fetched by ONE L2 instead of all eight
        streams = [by_clip[(which[by_clip] % 8) == xcd] for xcd in range(8)]
        blocks = min(len(stream) // 256 for stream in streams)
        interleaved = [stream[block * 256: (block + 1) * 256] for block in range(blocks) for stream in streams]
        interleaved = np.concatenate(interleaved + [stream[blocks * 256:] for stream in streams])      # (what is left over, still sorted by clip, at the end)
        assert interleaved.size == n
        patterns["256 clips as drawn, requests sorted by clip, clip c on XCD c % 8"] = tuple(column[interleaved] for column in drawn)
        library_order = runtime.order_track_requests_for_locality(crowd_handles[which].astype(np.uint32))
        patterns["256 clips as drawn, requests in aclhip_order_track_requests_for_locality order"] = tuple(column[library_order] for column in drawn)
        patterns["256 clips as drawn, root bone"] = (crowd_handles[which], (rng.uniform(0, 1, size=n).astype(np.float32) * crowd_durations[which]).astype(np.float32), np.zeros(n, dtype=np.int32))
        same_time = np.repeat(rng.uniform(0, one.duration, size=(n + 99) // 100).astype(np.float32), 100)[:n]
        patterns["one clip, all 100 bones of each instance"] = (np.full(n, one_handle, dtype=np.int32), same_time, (np.arange(n) % 100).astype(np.int32))
        only = os.environ.get("TRACK_SWEEP_ONLY")           # one pattern per process (counter passes)
        for name, (ids, times, tracks) in patterns.items():
            if only and name != only:
                continue
            d_ids, d_times, d_tracks = torch.from_numpy(ids).cuda(), torch.from_numpy(times).cuda(), torch.from_numpy(tracks).cuda()
            out = torch.empty((n, 12), dtype=torch.float32, device="cuda")
            ok = check_against_poses(ctx, d_ids, d_times, d_tracks, out, 100)
            us = time_requests(ctx, d_ids, d_times, d_tracks, out, repeats=int(os.environ.get("TRACK_SWEEP_REPEATS", "100")))
            print(f"decompress_track {n:8d} requests  {name:74s} {us:8.1f} us  {n / us / 1e3:7.2f} G bones/s  {n * 60 / us / 1e3:7.0f} GB/s algorithmic  {'== whole pose' if ok else 'DIFFERS FROM THE WHOLE POSE'}", flush=True)
    if ctx.rejected_instance_count() != 0:
        print("REJECTED", ctx.rejected_instance_count())
    ctx.close()


if __name__ == "__main__":
    main()
