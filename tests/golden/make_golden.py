#!/usr/bin/env python3
"""Generates tests/golden/*.npz: compressed clips (bytes) + sample times + the poses produced by the REFERENCE's own,
unmodified decoder (oracle/_ref/libaclref.so, built from /root/reference by oracle/Makefile).

Run in the build container (where /root/reference exists):  python tests/golden/make_golden.py
The fixtures are committed so that the oracle and the GPU path can be checked on machines without the reference.
Each case stores: blob, times, rounding policy, settings id (0 default_transform_decompression_settings, 1 debug = always
normalize + per track rounding, 2 default + per track rounding), default mode, optional defaults / per track policies,
poses [n, num_tracks, 12] from decompress_tracks and single [n, 12] from decompress_track(track_indices[i]).
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from acl_amd import synth  # noqa: E402
from oracle import bindings as ob  # noqa: E402

CASES = {
    # name: (clip spec, settings, default_mode, rounding policies)
    "cmu_70": (dict(seed=7, num_tracks=70, num_samples=301), 0, 0),
    "single_segment_scale": (dict(seed=21, num_tracks=24, num_samples=20, has_scale=1), 0, 0),
    "stripped_wrap_scale": (dict(seed=22, strip_keyframes=1, wrap=1, num_samples=100, has_scale=1, num_tracks=19), 0, 0),
    "v2_0_low_bits": (dict(seed=23, version=7, min_bits=3, max_bits=19, num_tracks=30, num_samples=90), 0, 0),
    "raw_and_constant_rates": (dict(seed=24, raw_fraction=0.3, width0_fraction=0.3, num_tracks=40, num_samples=70, has_scale=1, scale_default=0.2, scale_constant=0.2, translation_constant=0.3), 0, 0),
    "debug_settings_always_normalize": (dict(seed=25, num_tracks=33, num_samples=64, has_scale=1), 1, 0),
    "per_track_rounding": (dict(seed=26, num_tracks=33, num_samples=64, translation_constant=0.5), 2, 0),
    "variable_defaults": (dict(seed=27, num_tracks=20, num_samples=40, rotation_default=0.3, translation_default=0.3, has_scale=1, scale_default=0.6), 0, 3),
    "constant_defaults": (dict(seed=28, num_tracks=20, num_samples=40, rotation_default=0.3, translation_default=0.3), 0, 2),
    "skipped_defaults": (dict(seed=29, num_tracks=20, num_samples=40, rotation_default=0.3, translation_default=0.3, has_scale=1, scale_default=0.6), 0, 1),
}


# Clips compressed by the REFERENCE's own compressor (oracle/_ref/libaclref_compress.so: compress_track_list with
# get_default_compression_settings()) from synthetic raw animation: name -> (raw animation spec, compressor options)
REAL_COMPRESSOR_CASES = {
    "real_default_settings_70_bones": (dict(seed=41, num_tracks=70, num_samples=301), dict()),                    # BASELINE.json configs[0]
    "real_scale_single_segment": (dict(seed=42, num_tracks=24, num_samples=25, has_scale=1, scale_default=0.3), dict()),
    "real_keyframe_stripping": (dict(seed=43, num_tracks=30, num_samples=150), dict(strip_proportion=0.4)),
    "real_loop_optimized": (dict(seed=44, num_tracks=20, num_samples=91), dict(optimize_loops=True, make_looping=True)),
    "real_high_precision": (dict(seed=45, num_tracks=16, num_samples=64, translation_extent=50.0), dict(precision=0.000001)),
}


def real_compressor_blob(spec, options):
    options = dict(options)
    raw_clip = synth.build_clip(with_side_data=True, **spec)
    raw = raw_clip.raw_keyframes.copy()
    if options.pop("make_looping", False):
        raw[-1] = raw[0]        # first == last sample: the compressor drops the last one and sets the wrap flag
    return ob.ref_compress(raw, raw_clip.sample_rate, **options)


def main():
    if not ob.have_ref():
        raise SystemExit("oracle/_ref/libaclref.so is missing: run `make -C oracle ref` where /root/reference exists")
    rng = np.random.default_rng(2024)
    cases = dict(CASES)
    if ob.have_ref_compressor():
        for name, (spec, options) in REAL_COMPRESSOR_CASES.items():
            cases[name] = (real_compressor_blob(spec, options), 0, 0)
    for name, (spec, settings, default_mode) in cases.items():
        blob = spec if isinstance(spec, np.ndarray) else synth.build_clip(**spec).blob
        num_tracks = ob.ref().aclref_get_num_tracks(blob.ctypes.data)
        duration = ob.ref().aclref_get_duration(blob.ctypes.data, -1)
        times = np.concatenate([rng.uniform(-0.05, duration + 0.05, size=40), [0.0, duration, duration * 0.5]]).astype(np.float32)

        defaults = None
        if default_mode == 3:
            defaults = rng.uniform(-1.0, 1.0, size=(num_tracks, 12)).astype(np.float32)
        elif default_mode == 2:
            defaults = rng.uniform(-1.0, 1.0, size=(1, 12)).astype(np.float32)
        track_rounding = rng.integers(0, 4, size=num_tracks).astype(np.uint8) if settings in (1, 2) else None

        policies = [0, 1, 2, 3] + ([4] if track_rounding is not None else [])
        poses = np.zeros((len(policies), times.size, num_tracks, 12), dtype=np.float32)
        single = np.zeros((len(policies), times.size, 12), dtype=np.float32)
        track_indices = rng.integers(0, num_tracks, size=times.size).astype(np.uint32)
        prefill = rng.uniform(-5.0, 5.0, size=(num_tracks, 12)).astype(np.float32)

        for p, policy in enumerate(policies):
            for i, t in enumerate(times):
                out = prefill.copy()        # skipped defaults keep what was there
                ob.ref_decompress(blob, float(t), policy, -1, settings, default_mode, -1, defaults, track_rounding, out=out)
                out[:, 7] = 0.0             # W lanes of translation / scale are unspecified in the reference
                out[:, 11] = 0.0
                poses[p, i] = out
                one = prefill[track_indices[i]:track_indices[i] + 1].copy()
                # decompress_track writes to the track's slot of a full pose buffer: give it one
                full = prefill.copy()
                ob.ref_decompress(blob, float(t), policy, -1, settings, default_mode, int(track_indices[i]), defaults, track_rounding, out=full)
                one[0] = full[track_indices[i]]
                one[0, 7] = 0.0
                one[0, 11] = 0.0
                single[p, i] = one[0]

        path = os.path.join(HERE, f"{name}.npz")
        np.savez_compressed(path, blob=np.asarray(blob), times=times, policies=np.array(policies, dtype=np.uint8), settings=np.int32(settings),
                            default_mode=np.int32(default_mode), defaults=defaults if defaults is not None else np.zeros((0, 12), np.float32),
                            track_rounding=track_rounding if track_rounding is not None else np.zeros(0, np.uint8),
                            track_indices=track_indices, prefill=prefill, poses=poses, single=single)
        print(f"{name}: {os.path.getsize(path)} bytes, {num_tracks} tracks, {times.size} times x {len(policies)} policies")


if __name__ == "__main__":
    main()
