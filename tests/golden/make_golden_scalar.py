#!/usr/bin/env python3
"""Generates tests/golden/scalar/*.npz: scalar track lists (float1f / float2f / float3f / float4f / vector4f) compressed by the
REFERENCE's own compress_track_list and decoded by its own decompression_context (oracle/_ref/libaclref_scalar.so, built from
/root/reference by oracle/Makefile from oracle/ref_scalar_bridge.cpp).

Run in the build container (where /root/reference exists):  python tests/golden/make_golden_scalar.py
Stored per case: blob, times, policies (0..3 with default_scalar_decompression_settings, 4 = per_track with the debug settings and
`track_rounding`), values [len(policies), len(times), num_tracks, C] from decompress_tracks, track_indices + single
[len(policies), len(times), C] from decompress_track, and values_clamp / values_wrap with the looping policy forced.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import bindings as ob  # noqa: E402

CASES = {
    # name: (track type, tracks, samples, sample rate, precision, optimize loops / looping input)
    "float1f_blend_curves": (ob.TRACK_FLOAT1F, 40, 121, 30.0, 1e-4, False),
    "float2f_uv_scroll": (ob.TRACK_FLOAT2F, 9, 64, 24.0, 1e-3, False),
    "float3f_looping": (ob.TRACK_FLOAT3F, 17, 91, 30.0, 1e-4, True),
    "float4f_colors": (ob.TRACK_FLOAT4F, 12, 50, 60.0, 1e-5, False),
    "vector4f_wide_range": (ob.TRACK_VECTOR4F, 21, 33, 30.0, 1e-2, False),
}


def raw_samples(rng, num_samples, num_tracks, components, sample_rate, looping):
    t = np.arange(num_samples)[:, None, None] / sample_rate
    period = (num_samples - 1) / sample_rate
    frequency = rng.integers(1, 5, size=(1, num_tracks, components)) * (2.0 * np.pi / period)     # whole periods: first == last sample
    raw = np.sin(t * frequency + rng.uniform(0, 6, size=(1, num_tracks, components))) * rng.uniform(0.01, 30.0, size=(1, num_tracks, components))
    raw = raw.astype(np.float32)
    raw[:, 2] = raw[0, 2]                                                      # a constant track
    if num_tracks > 6:
        raw[:, 6] = rng.uniform(-1e5, 1e5, size=(num_samples, components))    # noise: ends up at the raw bit rate
        raw[:, 5] *= 1e-3                                                      # tiny range: few bits
    if looping:
        raw[-1] = raw[0]
    return raw


def main():
    if not ob.have_ref_scalar():
        raise SystemExit("oracle/_ref/libaclref_scalar.so is missing: run `make -C oracle ref` where /root/reference exists")
    rng = np.random.default_rng(4242)
    os.makedirs(os.path.join(HERE, "scalar"), exist_ok=True)
    for name, (track_type, num_tracks, num_samples, sample_rate, precision, looping) in CASES.items():
        components = {0: 1, 1: 2, 2: 3, 3: 4, 4: 4}[track_type]
        raw = raw_samples(rng, num_samples, num_tracks, components, sample_rate, looping)
        blob = ob.ref_scalar_compress(raw, track_type, sample_rate, precision=precision, optimize_loops=looping)
        duration = ob.ref().aclref_get_duration(blob.ctypes.data, -1)
        times = np.concatenate([rng.uniform(-0.05, duration + 0.05, size=30), [0.0, duration, duration * 0.5]]).astype(np.float32)
        track_rounding = rng.integers(0, 4, size=num_tracks).astype(np.uint8)
        track_indices = rng.integers(0, num_tracks, size=times.size).astype(np.uint32)
        policies = [0, 1, 2, 3, 4]
        values = np.zeros((len(policies), times.size, num_tracks, components), dtype=np.float32)
        single = np.zeros((len(policies), times.size, components), dtype=np.float32)
        looped = {0: np.zeros((times.size, num_tracks, components), np.float32), 1: np.zeros((times.size, num_tracks, components), np.float32)}
        for p, policy in enumerate(policies):
            settings = 1 if policy == 4 else 0
            for i, t in enumerate(times):
                values[p, i] = ob.ref_scalar_decompress(blob, float(t), policy, settings=settings, track_rounding=track_rounding)
                full = np.zeros((num_tracks, components), dtype=np.float32)
                ob.ref_scalar_decompress(blob, float(t), policy, settings=settings, track_index=int(track_indices[i]), track_rounding=track_rounding, out=full)
                single[p, i] = full[track_indices[i]]
        for looping_policy in (0, 1):
            for i, t in enumerate(times):
                looped[looping_policy][i] = ob.ref_scalar_decompress(blob, float(t), 0, looping=looping_policy)
        path = os.path.join(HERE, "scalar", f"{name}.npz")
        np.savez_compressed(path, blob=np.asarray(blob), times=times, policies=np.array(policies, dtype=np.uint8), track_rounding=track_rounding,
                            track_indices=track_indices, values=values, single=single, values_clamp=looped[0], values_wrap=looped[1], raw=raw)
        bit_rates = bytes(blob[32 + int(np.frombuffer(bytes(blob[36:40]), dtype=np.uint32)[0]):][:num_tracks])
        print(f"{name}: {os.path.getsize(path)} bytes, blob {blob.size} bytes, bit rates {sorted(set(bit_rates))}, duration {duration:.3f}")


if __name__ == "__main__":
    main()
