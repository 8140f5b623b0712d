"""The CPU oracle against the reference's own headers compiled into oracle/_ref (only where /root/reference was available
to build it; the committed golden fixtures cover machines without it). No GPU."""
import numpy as np
import pytest

from acl_amd import synth
from oracle import bindings as ob
import helpers
from conftest import CLIP_SPECS, sample_times_for

pytestmark = pytest.mark.skipif(not ob.have_ref(), reason="oracle/_ref/libaclref.so not built (needs /root/reference)")


@pytest.mark.parametrize("name", sorted(CLIP_SPECS))
def test_decompress_tracks_bit_exact_for_every_rounding_policy(name):
    clip = synth.build_clip(**CLIP_SPECS[name])
    assert ob.ref().aclref_is_valid(clip.blob.ctypes.data, 1) == 0
    duration = ob.ref().aclref_get_duration(clip.blob.ctypes.data, -1)
    assert duration == ob.oracle().aclo_finite_duration(clip.blob.ctypes.data, ob.LOOP_AS_COMPRESSED)
    rng = np.random.default_rng(1)
    times = sample_times_for(duration, 60, rng)
    for policy in (ob.ROUND_NONE, ob.ROUND_FLOOR, ob.ROUND_CEIL, ob.ROUND_NEAREST):
        for t in times:
            expected = ob.ref_decompress(clip.blob, float(t), policy)
            expected[:, 7] = 0.0
            expected[:, 11] = 0.0
            actual = ob.oracle_decompress_tracks(clip.blob, float(t), policy)
            assert helpers.bit_equal(actual, expected), f"{name} policy {policy} t {t}: {helpers.max_abs_diff(actual, expected)}"


@pytest.mark.parametrize("name", ["cmu_100", "stripped_wrap_scale", "raw_and_constant_rates", "v2_0_low_bits"])
def test_blobs_pass_the_reference_asserts(name):
    """The reference built with ACL_ON_ASSERT_THROW checks normalized/finite outputs and internal invariants."""
    if not ob.have_ref(asserting=True):
        pytest.skip("asserting reference build missing")
    clip = synth.build_clip(**CLIP_SPECS[name])
    duration = ob.ref().aclref_get_duration(clip.blob.ctypes.data, -1)
    for t in np.linspace(0.0, duration, 25, dtype=np.float32):
        ob.ref_decompress(clip.blob, float(t), ob.ROUND_NONE, settings=1, asserting=True)
        ob.ref_decompress(clip.blob, float(t), ob.ROUND_NONE, settings=0, track_index=0, asserting=True)


@pytest.mark.parametrize("looping", [ob.LOOP_CLAMP, ob.LOOP_WRAP])
@pytest.mark.parametrize("name", ["wrap_77", "cmu_70_default", "stripped_wrap_scale"])
def test_looping_policy_override(name, looping):
    clip = synth.build_clip(**CLIP_SPECS[name])
    duration = ob.ref().aclref_get_duration(clip.blob.ctypes.data, looping)
    assert duration == ob.oracle().aclo_finite_duration(clip.blob.ctypes.data, looping)
    options = helpers.oracle_options(looping=looping)
    rng = np.random.default_rng(2)
    for t in sample_times_for(duration, 40, rng):
        expected = ob.ref_decompress(clip.blob, float(t), ob.ROUND_NONE, looping=looping)
        expected[:, 7] = 0.0
        expected[:, 11] = 0.0
        actual = ob.oracle_decompress_tracks(clip.blob, float(t), ob.ROUND_NONE, options)
        assert helpers.bit_equal(actual, expected)


@pytest.mark.parametrize("settings", [1, 2])
@pytest.mark.parametrize("name", ["scale_37", "stripped", "raw_and_constant_rates"])
def test_normalization_and_per_track_rounding_settings(name, settings):
    clip = synth.build_clip(**CLIP_SPECS[name])
    duration = ob.ref().aclref_get_duration(clip.blob.ctypes.data, -1)
    rng = np.random.default_rng(3)
    track_rounding = rng.integers(0, 4, size=clip.num_tracks).astype(np.uint8)
    options = helpers.oracle_options(settings, 0, None, track_rounding)
    for policy in (0, 1, 2, 3, 4):
        for t in sample_times_for(duration, 25, rng):
            expected = ob.ref_decompress(clip.blob, float(t), policy, settings=settings, track_rounding=track_rounding)
            expected[:, 7] = 0.0
            expected[:, 11] = 0.0
            actual = ob.oracle_decompress_tracks(clip.blob, float(t), policy, options)
            assert helpers.bit_equal(actual, expected), f"{name} settings {settings} policy {policy}"


@pytest.mark.parametrize("default_mode", [0, 1, 2, 3])
def test_default_sub_track_modes(default_mode):
    clip = synth.build_clip(seed=5, num_tracks=30, num_samples=50, rotation_default=0.3, translation_default=0.3, has_scale=1, scale_default=0.6)
    duration = ob.ref().aclref_get_duration(clip.blob.ctypes.data, -1)
    rng = np.random.default_rng(4)
    defaults = rng.uniform(-1, 1, size=(clip.num_tracks if default_mode == 3 else 1, 12)).astype(np.float32)
    prefill = rng.uniform(-3, 3, size=(clip.num_tracks, 12)).astype(np.float32)
    options = helpers.oracle_options(0, default_mode, defaults)
    for t in sample_times_for(duration, 20, rng):
        expected = prefill.copy()
        ob.ref_decompress(clip.blob, float(t), ob.ROUND_NONE, default_mode=default_mode, defaults=defaults, out=expected)
        actual = prefill.copy()
        ob.oracle_decompress_tracks(clip.blob, float(t), ob.ROUND_NONE, options, out=actual)
        # skipped defaults leave the pre-filled values in place on both sides; W lanes of translation/scale are not compared
        assert helpers.bit_equal(actual, expected)


@pytest.mark.parametrize("name", ["cmu_100", "scale_37", "stripped_wrap_scale", "raw_and_constant_rates", "two_segments_32"])
def test_decompress_track_close_to_reference(name):
    clip = synth.build_clip(**CLIP_SPECS[name])
    duration = ob.ref().aclref_get_duration(clip.blob.ctypes.data, -1)
    rng = np.random.default_rng(5)
    worst = 0.0
    for t in sample_times_for(duration, 30, rng):
        for track in rng.integers(0, clip.num_tracks, size=6):
            full = np.zeros((clip.num_tracks, 12), dtype=np.float32)
            ob.ref_decompress(clip.blob, float(t), ob.ROUND_NONE, track_index=int(track), out=full)
            actual = ob.oracle_decompress_track(clip.blob, float(t), int(track))
            worst = max(worst, helpers.max_abs_diff(actual, full[track]))
    assert worst <= 1e-6


@pytest.mark.skipif(not ob.have_ref_database(), reason="oracle/_ref/libaclref_db.so not built (needs /root/reference)")
@pytest.mark.parametrize("max_chunk_size", [4096, 16384])
def test_database_streaming_states_bit_exact(max_chunk_size):
    """The restated database_context (oracle/database.py) walks the same stream_in / stream_out script as the reference's, holes
    included, and decoding through its runtime headers gives the reference's poses after every request."""
    from oracle.database import OracleDatabase
    blobs = []
    for seed, (num_tracks, num_samples) in enumerate([(30, 150), (20, 90), (25, 200), (40, 333)]):
        raw = synth.build_clip(seed=60 + seed, num_tracks=num_tracks, num_samples=num_samples, with_side_data=True)
        blobs.append(ob.ref_db_compress(raw.raw_keyframes, raw.sample_rate))
    reference = ob.ReferenceDatabase(blobs, max_chunk_size=max_chunk_size)
    database = OracleDatabase(reference.database, reference.bulk[1], reference.bulk[2])
    assert database.num_chunks == reference.num_chunks
    rng = np.random.default_rng(9)

    def check(label):
        for c, clip in enumerate(reference.clips):
            assert database.contains(clip)
            duration = ob.ref().aclref_get_duration(clip.ctypes.data, -1)
            for t in sample_times_for(duration, 12, rng):
                for policy in (ob.ROUND_NONE, ob.ROUND_FLOOR, ob.ROUND_CEIL, ob.ROUND_NEAREST):
                    expected = reference.decompress(c, float(t), policy)
                    actual = database.decompress_tracks(clip, float(t), policy)
                    assert helpers.bit_equal(actual, expected), f"{label} clip {c} t {t} policy {policy}"

    check("nothing resident")
    script = [(1, 1, True), (2, 2, True), (1, 1, False), (1, 2, True), (1, 100, True), (2, 1, False), (2, 100, True),
              (1, 100, False), (1, 100, True), (2, 100, False), (1, 100, False)]
    for tier, num_chunks, stream_in in script:
        result = reference.stream(tier, num_chunks, stream_in)
        moved = database.stream_in(tier, num_chunks) if stream_in else database.stream_out(tier, num_chunks)
        assert (result == 1) == (moved != 0)
        check(f"after {(tier, num_chunks, stream_in)}")
    reference.close()


def test_non_finite_sample_times():
    clip = synth.build_clip(**CLIP_SPECS["cmu_70_default"])
    for t in (float("nan"), float("inf"), float("-inf"), 1e30, -1e30, -0.0):
        for policy in (ob.ROUND_NONE, ob.ROUND_FLOOR, ob.ROUND_CEIL, ob.ROUND_NEAREST):
            expected = ob.ref_decompress(clip.blob, t, policy)
            assert helpers.bit_equal(ob.oracle_decompress_tracks(clip.blob, t, policy), expected), f"time {t} policy {policy}"
