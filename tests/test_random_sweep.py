"""Seeded random clip shapes (tests/conftest.py: random_clip_specs): the oracle against the reference's own decoder on CPU
(where oracle/_ref exists), the HIP kernels against the oracle on the GPU."""
import numpy as np
import pytest

from acl_amd import synth
from oracle import bindings as ob
import helpers
from conftest import random_clip_specs, sample_times_for

CPU_SPECS = random_clip_specs(60, seed=1)
GPU_SPECS = random_clip_specs(60, seed=2)


@pytest.mark.skipif(not ob.have_ref(), reason="oracle/_ref/libaclref.so not built (needs /root/reference)")
@pytest.mark.parametrize("index", range(len(CPU_SPECS)))
def test_oracle_matches_reference_on_random_shapes(index):
    clip = synth.build_clip(**CPU_SPECS[index])
    assert ob.ref().aclref_is_valid(clip.blob.ctypes.data, 1) == 0
    duration = ob.ref().aclref_get_duration(clip.blob.ctypes.data, -1)
    rng = np.random.default_rng(index)
    for t in sample_times_for(duration, 12, rng):
        for policy in (ob.ROUND_NONE, ob.ROUND_FLOOR, ob.ROUND_CEIL, ob.ROUND_NEAREST):
            expected = ob.ref_decompress(clip.blob, float(t), policy)
            actual = ob.oracle_decompress_tracks(clip.blob, float(t), policy)
            assert helpers.bit_equal(actual, expected), f"spec {CPU_SPECS[index]} policy {policy} t {t}"
        # the asserting build of the reference checks its internal invariants on the generated blob
        if ob.have_ref(asserting=True):
            ob.ref_decompress(clip.blob, float(t), ob.ROUND_NONE, asserting=True)


@pytest.mark.gpu
def test_kernels_match_oracle_on_random_shapes():
    import os
    from acl_amd import runtime
    for kernel in ("common_case", "any_settings"):
        if kernel == "any_settings":
            os.environ["ACLHIP_FORCE_GENERIC_KERNEL"] = "1"
        try:
            context = runtime.Context(0)
        finally:
            os.environ.pop("ACLHIP_FORCE_GENERIC_KERNEL", None)
        clips = [synth.build_clip(**spec) for spec in GPU_SPECS]
        handles = np.array([context.register_clip(clip.blob) for clip in clips], dtype=np.uint32)
        rng = np.random.default_rng(77)
        per_clip = 24
        which = np.repeat(np.arange(len(clips)), per_clip)
        times = np.concatenate([sample_times_for(clip.duration, per_clip - 5, rng) for clip in clips]).astype(np.float32)
        max_tracks = max(clip.num_tracks for clip in clips)
        for policy in (0, 1, 2, 3):
            poses = context.decompress_tracks(handles[which], times, params=runtime.default_params(rounding_policy=policy), num_tracks=max_tracks)
            for i in range(which.size):
                clip = clips[which[i]]
                expected = ob.oracle_decompress_tracks(clip.blob, float(times[i]), policy)
                assert helpers.bit_equal(poses[i, : clip.num_tracks], expected), f"{kernel}: spec {GPU_SPECS[which[i]]} policy {policy} t {times[i]}"
        tracks = np.array([rng.integers(0, clips[w].num_tracks) for w in which], dtype=np.uint32)
        single = context.decompress_track(handles[which], times, tracks)
        whole = context.decompress_tracks(handles[which], times, num_tracks=max_tracks)
        for i in range(which.size):
            assert helpers.bit_equal(single[i], whole[i, tracks[i]])
        assert context.rejected_instance_count() == 0
        context.close()


@pytest.mark.skipif(not ob.have_ref_compressor(), reason="oracle/_ref/libaclref_compress.so not built (needs /root/reference)")
@pytest.mark.parametrize("index", range(8))
def test_oracle_matches_reference_on_clips_from_the_reference_compressor(index):
    """Raw animation -> the reference's compress_track_list (default settings, some with keyframe stripping or loop optimisation)
    -> reference decoder vs oracle, bit for bit."""
    rng = np.random.default_rng(900 + index)
    num_tracks = int(rng.choice([5, 16, 33, 48]))
    num_samples = int(rng.choice([2, 17, 40, 95]))
    raw_clip = synth.build_clip(seed=900 + index, num_tracks=num_tracks, num_samples=num_samples, has_scale=int(index % 2), scale_default=0.5,
                                with_side_data=True)
    raw = raw_clip.raw_keyframes.copy()
    options = [dict(), dict(strip_proportion=0.3), dict(optimize_loops=True), dict(precision=0.001)][index % 4]
    if options.get("optimize_loops"):
        raw[-1] = raw[0]
    blob = ob.ref_compress(raw, raw_clip.sample_rate, **options)
    assert ob.ref().aclref_is_valid(blob.ctypes.data, 1) == 0
    duration = ob.ref().aclref_get_duration(blob.ctypes.data, -1)
    for t in sample_times_for(duration, 10, rng):
        for policy in (ob.ROUND_NONE, ob.ROUND_FLOOR, ob.ROUND_CEIL, ob.ROUND_NEAREST):
            expected = ob.ref_decompress(blob, float(t), policy)
            actual = ob.oracle_decompress_tracks(blob, float(t), policy)
            assert helpers.bit_equal(actual, expected), f"clip {index} policy {policy} t {t}"
