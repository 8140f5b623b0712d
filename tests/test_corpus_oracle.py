"""The real-compressor corpus (tests/golden/corpus, written by the reference's own compress_track_list / build_database through
tests/golden/make_corpus.py) on the CPU side: every blob passes the product's registration-time validation -- none of the refusals
that are stricter than the reference's is_valid() fires on anything the reference's compressor writes --, and the C oracle equals the
reference's own decoder (oracle/_ref, where it exists) on every sample of every clip. No GPU.

Mirrors tools/acl_compressor/sources/validate_tracks.cpp:92-260 (validate_accuracy): debug_transform_decompression_settings, every
sample of the clip at min(i / rate, duration), rounding policy nearest."""
import numpy as np
import pytest

from acl_amd import runtime
from oracle import bindings as ob
import helpers

CORPUS = helpers.load_corpus()
NAMES = [clip["name"] for clip in CORPUS]


def test_corpus_spans_what_the_reference_regression_set_spans():
    specs = [clip["spec"] for clip in CORPUS]
    assert len(CORPUS) >= 200
    configs = {spec["config"] for spec in specs}
    assert {"quant_medium", "quant_high", "quant_highest", "raw", "mixed_var_0", "mixed_var_1", "quant_mtx_error", "quant_bind_relative", "keyframe_stripping"} <= configs
    assert {1, 2, 16, 17, 31, 32, 33, 600} <= {spec["samples"] for spec in specs}
    assert {1, 2, 3, 4, 5, 16, 17, 104, 105, 300, 551} <= {spec["bones"] for spec in specs}
    assert any(spec.get("looping") for spec in specs) and any(spec.get("scale") for spec in specs) and any(spec.get("mirrored") for spec in specs)
    assert sum(clip["blob"].size for clip in CORPUS) < 10 * 1024 * 1024


def test_every_corpus_blob_passes_registration_time_validation():
    """aclhip_check_clip = everything aclhip_register_clip does on the host (validate_clip + the table derivation with its refusals):
    0 of the corpus' blobs are refused -- databases' clips included"""
    refused = []
    for clip in CORPUS:
        status, message = runtime.check_clip(clip["blob"])
        if status != 0:
            refused.append((clip["name"], status, message))
    for name in helpers.CORPUS_DATABASES:
        case = helpers.load_corpus_database(name)
        for index, blob in enumerate(case["clips"]):
            status, message = runtime.check_clip(blob)
            if status != 0:
                refused.append((f"{name}[{index}]", status, message))
        status, message = runtime.check_database(case["database"], case["bulk_medium"] if case["bulk_medium"].size else None, case["bulk_low"] if case["bulk_low"].size else None)
        if status != 0:
            refused.append((name, status, message))
    assert refused == []


def _formats_of(blob):
    misc_packed = int(np.frombuffer(bytes(blob[28:32]), dtype=np.uint32)[0])
    return (misc_packed >> 4) & 15, (misc_packed >> 3) & 1, (misc_packed >> 2) & 1


def test_corpus_holds_every_packed_format():
    seen = {_formats_of(clip["blob"]) for clip in CORPUS if np.frombuffer(bytes(clip["blob"][16:20]), dtype=np.uint32)[0] != 0}
    assert {(0, 0, 0), (0, 1, 1), (3, 0, 1), (2, 1, 0), (3, 1, 1)} <= seen


@pytest.mark.skipif(not ob.have_ref(), reason="oracle/_ref/libaclref.so not built (needs /root/reference)")
@pytest.mark.parametrize("name", NAMES)
def test_oracle_equals_the_reference_on_every_sample(name):
    clip = CORPUS[NAMES.index(name)]
    blob = clip["blob"]
    assert ob.ref().aclref_is_valid(blob.ctypes.data, 1) == 0
    num_tracks = ob.oracle().aclo_num_tracks(blob.ctypes.data)
    times, duration = helpers.corpus_sample_times(blob)
    assert duration == ob.ref().aclref_get_duration(blob.ctypes.data, -1)
    # debug settings (normalize always, per track rounding: what validate_accuracy decodes with) for every clip; the default
    # settings where the reference's default settings accept the clip (variable formats), "default + every format" otherwise
    variable_only = _formats_of(blob) == (3, 1, 1) or num_tracks == 0
    track_rounding = np.full(max(num_tracks, 1), ob.ROUND_NEAREST, dtype=np.uint8)
    for settings in (1, 0 if variable_only else 4):
        options = helpers.oracle_options(settings, 0, None, track_rounding if settings == 1 else None)
        for policy in ((ob.ROUND_NEAREST, ob.ROUND_PER_TRACK) if settings == 1 else (ob.ROUND_NEAREST, ob.ROUND_NONE)):
            for t in list(times) + ([float(times[len(times) // 2]) + 0.4 / float(ob.oracle().aclo_sample_rate(blob.ctypes.data))] if len(times) else []):
                expected = ob.ref_decompress(blob, float(t), policy, settings=settings, track_rounding=track_rounding)
                expected[:, 7] = 0.0
                expected[:, 11] = 0.0
                actual = ob.oracle_decompress_tracks(blob, float(t), policy, options)
                assert helpers.bit_equal(actual, expected), f"{name}: settings {settings} policy {policy} t {t}: {helpers.max_abs_diff(actual, expected)}"


@pytest.mark.skipif(not ob.have_ref(), reason="oracle/_ref/libaclref.so not built (needs /root/reference)")
@pytest.mark.parametrize("name", NAMES[::4])
def test_oracle_single_track_is_the_reference_single_track(name):
    """decompress_track for every bone at a few samples: within 1e-6 of the reference's (its rotation takes rtm::quat_lerp, whose
    reciprocal square root the shim restates: oracle/rtm_shim) and bit identical for translations and scales"""
    clip = CORPUS[NAMES.index(name)]
    blob = clip["blob"]
    num_tracks = ob.oracle().aclo_num_tracks(blob.ctypes.data)
    times, _ = helpers.corpus_sample_times(blob)
    variable_only = _formats_of(blob) == (3, 1, 1) or num_tracks == 0
    settings = 0 if variable_only else 4
    options = helpers.oracle_options(settings)
    for t in times[:: max(1, len(times) // 5)]:
        for track in range(0, num_tracks, max(1, num_tracks // 40)):
            full = np.zeros((num_tracks, 12), dtype=np.float32)
            ob.ref_decompress(blob, float(t), ob.ROUND_NONE, settings=settings, track_index=track, out=full)
            actual = ob.oracle_decompress_track(blob, float(t), track, ob.ROUND_NONE, options)
            assert helpers.max_abs_diff(actual, full[track]) <= 1e-6
            assert np.array_equal(actual[[4, 5, 6, 8, 9, 10]].view(np.uint32), full[track][[4, 5, 6, 8, 9, 10]].view(np.uint32))


def test_optional_metadata_is_read_like_the_reference_reads_it():
    """aclhip_read_clip_metadata (host only) against the corpus' own sources -- the skeleton and bind pose every clip was compressed with --
    and, where oracle/_ref exists, against compressed_tracks::get_parent_track_index / get_track_description themselves
    (core/impl/compressed_tracks.impl.h:175-275)"""
    with_parents = with_descriptions = 0
    for clip in CORPUS:
        spec, blob = clip["spec"], clip["blob"]
        info, parents, descriptions = runtime.read_clip_metadata(blob)
        wants_descriptions = bool(spec.get("include_track_descriptions"))
        wants_parents = wants_descriptions or bool(spec.get("include_parent_track_indices"))       # (descriptions bring the parents along, compress.transform.impl.h:182-183)
        assert bool(info.has_parent_track_indices) == wants_parents and bool(info.has_track_descriptions) == wants_descriptions, clip["name"]
        assert bool(info.has_track_names) == bool(spec.get("include_track_names")) and bool(info.has_track_list_name) == bool(spec.get("include_track_list_name"))
        if wants_parents:
            with_parents += 1
            assert np.array_equal(parents, clip["parents"].astype(np.uint32))                        # (-1 = 0xFFFFFFFF = k_invalid_track_index)
        if wants_descriptions:
            with_descriptions += 1
            defaults, precisions, shells = descriptions
            expected = clip["bind_pose"].copy() if clip["bind_is_default"] else np.tile(np.array([0, 0, 0, 1, 0, 0, 0, 0, 1, 1, 1, 0], dtype=np.float32), (spec["bones"], 1))
            expected[:, 7] = 0.0
            expected[:, 11] = 0.0
            assert np.array_equal(defaults.view(np.uint32), expected.view(np.uint32)), clip["name"]
            assert np.all(precisions == np.float32(spec.get("precision", 0.01))) and np.all(shells == np.float32(spec.get("shell_distance", 3.0)))
        if ob.have_ref() and hasattr(ob.ref(), "aclref_get_metadata"):
            ref_parents, ref_descriptions = ob.ref_get_metadata(blob)
            assert (ref_descriptions is not None) == wants_descriptions
            if wants_parents:
                assert np.array_equal(parents, ref_parents)
            else:
                assert np.all(ref_parents == 0xFFFFFFFF)
            if wants_descriptions:
                for ours, theirs in zip(descriptions, ref_descriptions):
                    assert np.array_equal(ours.view(np.uint32), theirs.view(np.uint32)), clip["name"]
    assert with_parents >= 3 and with_descriptions >= 3


def test_metadata_offsets_that_leave_the_blob_count_as_not_stored():
    clip = next(clip for clip in CORPUS if clip["spec"].get("include_track_descriptions") and clip["spec"].get("include_track_names"))
    blob = clip["blob"]
    for field in range(5):
        for value in (blob.size - 8, 0x7FFFFFF0, 0xFFFFFFFE):
            broken = blob.copy()
            broken[blob.size - 20 + 4 * field: blob.size - 16 + 4 * field] = np.frombuffer(np.uint32(value).tobytes(), dtype=np.uint8)
            info, parents, descriptions = runtime.read_clip_metadata(broken)
            stored = [info.has_track_list_name, info.has_track_names, info.has_parent_track_indices, info.has_track_descriptions, info.has_contributing_error]
            assert info.has_metadata == 1 and (stored[field] == 0 or value == blob.size - 8 and field in (0, 4))
            if field == 2:
                assert parents is None and descriptions is None       # (descriptions come with the parent indices, compressed_tracks.impl.h:229-230)
