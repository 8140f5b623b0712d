"""Scalar track lists (float1f / float2f / float3f / float4f / vector4f): the CPU oracle against golden vectors from the reference's
own compressor + decoder (tests/golden/scalar/*.npz), against the synthetic writer's predictions, and -- where oracle/_ref was
built -- against the reference live. No GPU."""
import numpy as np
import pytest

from acl_amd import synth
from oracle import bindings as ob
import helpers


@pytest.mark.parametrize("name", helpers.scalar_golden_cases())
def test_oracle_matches_reference_golden(name):
    case = helpers.load_scalar_golden(name)
    blob = case["blob"]
    for p, policy in enumerate(case["policies"]):
        options = ob.default_options()
        if policy == ob.ROUND_PER_TRACK:
            options = ob.default_options(per_track_rounding=1, track_rounding=case["track_rounding"].ctypes.data)
        for i, t in enumerate(case["times"]):
            values = ob.oracle_scalar_decompress_tracks(blob, float(t), int(policy), options)
            assert helpers.exact(values, case["values"][p, i]), f"{name}: policy {policy} time {t}"
            single = ob.oracle_scalar_decompress_track(blob, float(t), int(case["track_indices"][i]), int(policy), options)
            assert helpers.exact(single, case["single"][p, i])
    for looping, key in ((ob.LOOP_CLAMP, "values_clamp"), (ob.LOOP_WRAP, "values_wrap")):
        for i, t in enumerate(case["times"]):
            values = ob.oracle_scalar_decompress_tracks(blob, float(t), ob.ROUND_NONE, ob.default_options(looping_policy=looping))
            assert helpers.exact(values, case[key][i])


@pytest.mark.parametrize("name", helpers.scalar_golden_cases())
def test_reference_compressor_honours_its_precision(name):
    """Decoding at the sample times reproduces the raw input within the precision the clip was compressed with."""
    case = helpers.load_scalar_golden(name)
    raw = case["raw"]
    num_samples = ob.oracle().aclo_num_samples(case["blob"].ctypes.data)
    rate = ob.oracle().aclo_sample_rate(case["blob"].ctypes.data)
    precision = {"float1f_blend_curves": 1e-4, "float2f_uv_scroll": 1e-3, "float3f_looping": 1e-4, "float4f_colors": 1e-5, "vector4f_wide_range": 1e-2}[name]
    for sample in range(0, num_samples, 7):
        values = ob.oracle_scalar_decompress_tracks(case["blob"], sample / rate, ob.ROUND_NEAREST)
        assert np.abs(values - raw[sample]).max() <= precision * 1.01 + np.abs(raw[sample]).max() * 1e-6


@pytest.mark.parametrize("name", sorted(helpers.SCALAR_CLIP_SPECS))
def test_synthetic_writer_predicts_what_the_oracle_decodes(name):
    clip = synth.build_scalar_clip(**helpers.SCALAR_CLIP_SPECS[name])
    assert ob.oracle().aclo_hash32(clip.blob[8:].ctypes.data, clip.blob.size - 8) == int(np.frombuffer(bytes(clip.blob[4:8]), dtype=np.uint32)[0])
    assert ob.oracle().aclo_scalar_num_components(clip.blob.ctypes.data) == clip.num_components
    assert ob.oracle().aclo_finite_duration(clip.blob.ctypes.data, ob.LOOP_AS_COMPRESSED) == np.float32(clip.duration)
    for sample in range(clip.num_samples):
        values = ob.oracle_scalar_decompress_tracks(clip.blob, sample / clip.sample_rate, ob.ROUND_NEAREST)
        assert helpers.exact(values, clip.keyframes[sample]), f"{name}: sample {sample}"
    # interpolation: floor / ceil bracket none, nearest is one of them
    rng = np.random.default_rng(7)
    for t in rng.uniform(0.0, clip.duration, size=16) if clip.duration > 0.0 else []:
        lo = ob.oracle_scalar_decompress_tracks(clip.blob, float(t), ob.ROUND_FLOOR)
        hi = ob.oracle_scalar_decompress_tracks(clip.blob, float(t), ob.ROUND_CEIL)
        mid = ob.oracle_scalar_decompress_tracks(clip.blob, float(t), ob.ROUND_NONE)
        near = ob.oracle_scalar_decompress_tracks(clip.blob, float(t), ob.ROUND_NEAREST)
        tolerance = 1e-5 * (1.0 + np.maximum(np.abs(lo), np.abs(hi)))
        assert (mid >= np.minimum(lo, hi) - tolerance).all() and (mid <= np.maximum(lo, hi) + tolerance).all()
        assert (helpers.exact(near, lo) or helpers.exact(near, hi))


@pytest.mark.skipif(not ob.have_ref_scalar(), reason="oracle/_ref/libaclref_scalar.so not built (needs /root/reference)")
@pytest.mark.parametrize("name", sorted(helpers.SCALAR_CLIP_SPECS))
def test_synthetic_clips_bit_exact_against_the_reference(name):
    clip = synth.build_scalar_clip(**helpers.SCALAR_CLIP_SPECS[name])
    assert ob.ref().aclref_is_valid(clip.blob.ctypes.data, 1) == 0
    assert ob.ref().aclref_get_duration(clip.blob.ctypes.data, -1) == np.float32(clip.duration)
    rng = np.random.default_rng(8)
    times = np.concatenate([rng.uniform(-0.1, clip.duration + 0.1, size=25), [0.0, clip.duration]])
    policies = rng.integers(0, 4, size=clip.num_tracks).astype(np.uint8)
    for t in times:
        for policy in (ob.ROUND_NONE, ob.ROUND_FLOOR, ob.ROUND_CEIL, ob.ROUND_NEAREST):
            assert helpers.exact(ob.oracle_scalar_decompress_tracks(clip.blob, float(t), policy), ob.ref_scalar_decompress(clip.blob, float(t), policy))
        expected = ob.ref_scalar_decompress(clip.blob, float(t), ob.ROUND_PER_TRACK, settings=1, track_rounding=policies)
        actual = ob.oracle_scalar_decompress_tracks(clip.blob, float(t), ob.ROUND_PER_TRACK, ob.default_options(per_track_rounding=1, track_rounding=policies.ctypes.data))
        assert helpers.exact(actual, expected)
        for looping in (ob.LOOP_CLAMP, ob.LOOP_WRAP):
            expected = ob.ref_scalar_decompress(clip.blob, float(t), ob.ROUND_NONE, looping=looping)
            assert helpers.exact(ob.oracle_scalar_decompress_tracks(clip.blob, float(t), ob.ROUND_NONE, ob.default_options(looping_policy=looping)), expected)
        track = int(rng.integers(0, clip.num_tracks))
        full = np.zeros((clip.num_tracks, clip.num_components), dtype=np.float32)
        ob.ref_scalar_decompress(clip.blob, float(t), ob.ROUND_NONE, track_index=track, out=full)
        assert helpers.exact(ob.oracle_scalar_decompress_track(clip.blob, float(t), track), full[track])
