"""aclhip_check_clip / aclhip_check_database: the host side of registration (validation + table derivation) without a device.
What the reference's is_valid() accepts here must be accepted, broken buffers must be refused -- never crash. No GPU."""
import struct

import numpy as np
import pytest

from acl_amd import runtime, synth
import helpers
from conftest import CLIP_SPECS, random_clip_specs


@pytest.mark.parametrize("name", sorted(CLIP_SPECS))
def test_synthetic_transform_clips_are_valid(name):
    clip = synth.build_clip(**CLIP_SPECS[name])
    assert runtime.check_clip(clip.blob) == (0, "")


@pytest.mark.parametrize("name", sorted(helpers.SCALAR_CLIP_SPECS))
def test_synthetic_scalar_clips_are_valid(name):
    clip = synth.build_scalar_clip(**helpers.SCALAR_CLIP_SPECS[name])
    assert runtime.check_clip(clip.blob) == (0, "")


def test_reference_written_blobs_are_valid():
    for name in helpers.golden_cases():
        assert runtime.check_clip(helpers.load_golden(name)["blob"]) == (0, ""), name
    for name in helpers.scalar_golden_cases():
        assert runtime.check_clip(helpers.load_scalar_golden(name)["blob"]) == (0, ""), name
    for name in helpers.database_golden_cases():
        case = helpers.load_database_golden(name)
        for clip in case["clips"]:
            assert runtime.check_clip(clip) == (0, ""), name
        assert runtime.check_database(case["database"], case["bulk_medium"], case["bulk_low"]) == (0, "")
        assert runtime.check_database(case["database_inline"]) == (0, "")


def _patched(blob, offset, fmt, value):
    out = synth.aligned_bytes(blob.size)
    out[:] = blob
    struct.pack_into(fmt, out, offset, value)
    return out


def test_targeted_corruptions_are_refused_with_a_reason():
    clip = synth.build_clip(seed=3, num_tracks=37, num_samples=80, has_scale=1)
    blob = clip.blob
    assert runtime.check_clip(_patched(blob, 8, "<I", 0xDEADBEEF), check_hash=False) == (2, "Invalid tag")
    assert runtime.check_clip(_patched(blob, 12, "<H", 3), check_hash=False)[1] == "Invalid algorithm version"
    assert runtime.check_clip(_patched(blob, 14, "<B", 9), check_hash=False)[1] == "Invalid algorithm type"
    assert runtime.check_clip(_patched(blob, 0, "<I", blob.size + 1000), check_hash=False)[1] == "Invalid size"
    assert runtime.check_clip(_patched(blob, 40, "<I", 1))[1] == "Invalid hash"
    status, message = runtime.check_clip(_patched(blob, 15, "<B", 7), check_hash=False)         # track type: not qvvf, not scalar
    assert status == 3 and "track type" in message
    # formats rewritten to quatf_full + vector3f_full on a clip laid out for the variable ones: the sections no longer add up (since round 6
    # the full formats themselves register: tests/test_gpu_corpus.py)
    assert runtime.check_clip(_patched(blob, 28, "<I", 0), check_hash=False)[0] == 2
    misc_packed = int(np.frombuffer(bytes(blob[28:32]), dtype=np.uint32)[0])
    status, message = runtime.check_clip(_patched(blob, 28, "<I", (misc_packed & ~0xF0) | (1 << 4)), check_hash=False)      # rotation_format8 1: no such format
    assert status == 3 and "rotation format" in message
    assert runtime.check_clip(_patched(blob, 32, "<I", 0), check_hash=False)[0] != 0            # no segments
    assert runtime.check_clip(_patched(blob, 32 + 36, "<I", 0x7FFFFFF0), check_hash=False)[0] != 0      # segment headers offset
    assert runtime.check_clip(_patched(blob, 32 + 4, "<I", 9999), check_hash=False)[0] != 0     # animated sub-track count
    assert runtime.check_clip(blob[: blob.size // 2].copy(), check_hash=False)[0] != 0          # truncated

    curves = synth.build_scalar_clip(seed=5, track_type=1, num_tracks=10, num_samples=12)
    assert runtime.check_clip(_patched(curves.blob, 32, "<I", 5), check_hash=False)[0] != 0     # bits per frame
    assert runtime.check_clip(_patched(curves.blob, 32 + 16, "<I", 0x0FFFFFFF), check_hash=False)[0] != 0   # animated values offset
    assert runtime.check_clip(_patched(curves.blob, 52, "<B", 99), check_hash=False)[0] != 0    # first bit rate


def test_segment_offsets_near_uint32_max_do_not_wrap_into_the_buffer():
    """segment_header::segment_data is an untrusted 32 bit offset: values near 2^32 used to wrap past the bounds check
     and the registration then read 4 GiB out of bounds."""
    for name in ("raw_and_constant_rates", "cmu_100", "single_segment", "stripped", "cinematic_300"):
        clip = synth.build_clip(**CLIP_SPECS[name])
        blob = clip.blob
        transform_header = 32
        num_segments, = struct.unpack_from("<I", blob, transform_header)
        segment_headers_offset, = struct.unpack_from("<I", blob, transform_header + 36)
        first = transform_header + segment_headers_offset
        for segment in range(min(num_segments, 3)):
            for size in (16, 20):
                at = first + segment * size + 12        # segment_data is the last u32 of segment_header (both variants start with it)
                if at + 4 > blob.size:
                    continue
                for value in (0xFFFFFFDF, 0xFFFFFFFF, 0xFFFFFF00, 0x80000000, blob.size, blob.size - 1):
                    status, message = runtime.check_clip(_patched(blob, at, "<I", value), check_hash=False)
                    # a patched field that is not segment_data (wrong header size guess) may leave a valid clip: never a crash
                    assert status in (0, 2, 3), (name, segment, size, hex(value), message)
        # the exact field of the first segment: it must be refused
        offset_of_segment_data = first + 12
        for value in (0xFFFFFFDF, 0xFFFFFFFF, 0xFFFFFF00, 0x80000000):
            status, message = runtime.check_clip(_patched(blob, offset_of_segment_data, "<I", value), check_hash=False)
            assert status == 2 and "outside" in message, (name, hex(value), message)


def test_database_corruptions_are_refused():
    case = helpers.load_database_golden("three_clips_4k_chunks")
    database, medium, low = case["database"], case["bulk_medium"], case["bulk_low"]
    assert runtime.check_database(database)[0] != 0                                             # bulk data missing
    assert runtime.check_database(_patched(database, 8, "<I", 1), medium, low, check_hash=False)[0] != 0       # tag
    bad = medium.copy()
    bad[5] ^= 0xFF
    assert runtime.check_database(database, bad, low)[0] != 0                                   # bulk hash
    assert runtime.check_database(database, bad, low, check_hash=False)[0] != 0                 # chunk header now inconsistent (index / size)
    assert runtime.check_database(_patched(database, 8 + 8, "<I", 4000), medium, low, check_hash=False)[0] != 0     # number of chunks
    assert runtime.check_database(database[:30].copy(), medium, low, check_hash=False)[0] != 0


@pytest.mark.parametrize("seed", range(6))
def test_random_corruptions_never_crash_the_host_side(seed):
    rng = np.random.default_rng(seed)
    specs = random_clip_specs(6, seed=50 + seed)
    blobs = [synth.build_clip(**spec).blob for spec in specs]
    blobs += [synth.build_scalar_clip(**spec).blob for spec in list(helpers.SCALAR_CLIP_SPECS.values())[seed: seed + 2]]
    accepted = 0
    for blob in blobs:
        header_bytes = min(blob.size, 32 + 52 + 4 * 40 + 64)
        for _ in range(250):
            corrupt = synth.aligned_bytes(blob.size)
            corrupt[:] = blob
            for _ in range(int(rng.integers(1, 4))):
                # mostly in the headers, where every field steers an offset or a count
                at = int(rng.integers(0, header_bytes)) if rng.uniform() < 0.8 else int(rng.integers(0, blob.size))
                corrupt[at] = int(rng.integers(0, 256))
            size = blob.size if rng.uniform() < 0.9 else int(rng.integers(0, blob.size))
            status, _ = runtime.check_clip(corrupt[:size] if size else corrupt[:1], check_hash=False)
            accepted += int(status == 0)
    assert accepted < len(blobs) * 250          # (some corruptions only touch values, those are fine)


@pytest.mark.parametrize("seed", range(2))
def test_random_database_corruptions_never_crash_the_host_side(seed):
    rng = np.random.default_rng(100 + seed)
    for name in helpers.database_golden_cases():
        case = helpers.load_database_golden(name)
        for inline in (False, True):
            original = case["database_inline"] if inline else case["database"]
            for _ in range(150):
                database = synth.aligned_bytes(original.size)
                database[:] = original
                medium, low = case["bulk_medium"].copy(), case["bulk_low"].copy()
                for _ in range(int(rng.integers(1, 4))):
                    target = int(rng.integers(0, 3))
                    if target == 0 or inline:
                        at = int(rng.integers(0, min(database.size, 200))) if rng.uniform() < 0.7 else int(rng.integers(0, database.size))
                        database[at] = int(rng.integers(0, 256))
                    elif target == 1 and medium.size:
                        medium[int(rng.integers(0, min(medium.size, 400)))] = int(rng.integers(0, 256))
                    elif low.size:
                        low[int(rng.integers(0, min(low.size, 400)))] = int(rng.integers(0, 256))
                if inline:
                    runtime.check_database(database, check_hash=False)
                else:
                    runtime.check_database(database, medium if medium.size else None, low if low.size else None, check_hash=False)


def test_a_track_count_near_2_to_the_32_is_refused():
    """0xFFFFFFFF tracks: (num_tracks + 15) / 16 sub-track type words wrapped to zero in 32 bit arithmetic, the bounds test passed and
    registration went on to size its tables for four billion tracks (found with tools/fuzz_host_validators.py under AddressSanitizer)"""
    clip = synth.build_clip(seed=3, num_tracks=20, num_samples=20)
    for num_tracks in (0xFFFFFFFF, 0xFFFFFFF1, 0xFFFFFFF0, 0x80000000, 0x10000000):
        blob = clip.blob.copy()
        aligned = synth.aligned_bytes(blob.size)
        aligned[:] = blob
        aligned[16:20].view(np.uint32)[0] = num_tracks          # tracks_header::num_tracks (core/impl/compressed_headers.h)
        status, message = runtime.check_clip(aligned, check_hash=False)
        assert status != 0, (hex(num_tracks), message)


def test_mutated_buffers_are_refused_or_accepted_never_worse():
    """a slice of tools/fuzz_host_validators.py (fixed seed, 3 seconds): clips, inline databases, tier stripping, walk planning"""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    result = subprocess.run([sys.executable, os.path.join(root, "tools", "fuzz_host_validators.py"), "5", "3"], capture_output=True, text=True, timeout=300)
    assert result.returncode == 0 and "host validator fuzz ok" in result.stdout, result.stdout[-2000:] + result.stderr[-2000:]


TRANSFORM_HEADER_OFFSET = 32            # acl_format.h: k_transform_header_offset
SEGMENT_START_INDICES_OFFSET = 52       # acl_format.h: k_segment_start_indices_offset (relative to the transform header)


def segment_starts(blob):
    """the clip's own segment_start_indices (multi segment clips), as a writable view"""
    num_segments = int(np.frombuffer(blob, dtype=np.uint32, count=1, offset=TRANSFORM_HEADER_OFFSET)[0])
    assert num_segments > 1
    return blob[TRANSFORM_HEADER_OFFSET + SEGMENT_START_INDICES_OFFSET:][: 4 * num_segments].view(np.uint32)


def with_moved_start(clip, segment, delta):
    """the same blob with one segment start moved: a cut the compressor never makes (the bytes stay a valid clip: the keyframes the
    moved segment now claims lie inside the buffer; what they decode to is whatever bytes are there -- the same for every decoder)"""
    blob = clip.blob.copy()
    aligned = synth.aligned_bytes(blob.size)
    aligned[:] = blob
    starts = segment_starts(aligned)
    starts[segment] = int(starts[segment]) + delta
    return aligned


def test_segment_starts_moved_by_hand_are_refused():
    """a segment start moved by one sample makes one of its neighbours claim a keyframe it does not store (what lies there is the next
    segment's format data): the reference, its restatement and the kernels each decode something else from such a blob (measured with
    tools/fuzz_gpu_mutated.py) -- registration refuses it; so it does a list of starts without its 0xFFFFFFFF end, or one the reference's
    guess-and-scan lookup would not resolve to the same segments"""
    clip = synth.build_clip(seed=5, num_tracks=24, num_samples=100)          # 17 17 17 17 16 16
    assert runtime.check_clip(clip.blob)[0] == 0
    for segment, delta in ((1, -1), (1, 1), (2, 1), (4, -1), (5, 1), (5, -2), (3, 4)):
        status, message = runtime.check_clip(with_moved_start(clip, segment, delta), check_hash=False)
        assert status != 0, (segment, delta)
    unterminated = synth.aligned_bytes(clip.blob.size)
    unterminated[:] = clip.blob
    end = unterminated[TRANSFORM_HEADER_OFFSET + SEGMENT_START_INDICES_OFFSET + 4 * 6:][:4].view(np.uint32)     # the entry behind the last segment
    assert end[0] == 0xFFFFFFFF
    end[0] = 100
    assert runtime.check_clip(unterminated, check_hash=False)[0] != 0


def test_stripped_segments_keep_their_first_and_last_keyframe():
    clip = synth.build_clip(seed=9, num_tracks=12, num_samples=70, strip_keyframes=1)
    assert runtime.check_clip(clip.blob)[0] == 0
    num_segments = int(np.frombuffer(clip.blob, dtype=np.uint32, count=1, offset=32)[0])
    headers_offset = 32 + int(np.frombuffer(clip.blob, dtype=np.uint32, count=1, offset=32 + 36)[0])      # transform_tracks_header::segment_headers_offset
    for segment in range(num_segments):
        for clear in (0x80000000, None):
            blob = synth.aligned_bytes(clip.blob.size)
            blob[:] = clip.blob
            word = blob[headers_offset + segment * 20 + 16:][:4].view(np.uint32)        # stripped_segment_header::sample_indices
            indices = int(word[0])
            lowest = indices & -indices
            word[0] = indices & ~(clear if clear is not None else lowest)                # first / last kept keyframe gone
            status, _ = runtime.check_clip(blob, check_hash=False)
            assert status != 0, (segment, hex(indices))


def test_more_sub_tracks_of_a_kind_than_tracks_are_refused():
    """a full format's sub-track count appears in no sum the validator checked: 0xFFFFFFFF animated rotations (+ 2 translations) wrapped
    the table sizes at registration (found by tools/fuzz_gpu_mutated.py with full-format corpus sources under AddressSanitizer, round 6)"""
    raw = [clip for clip in helpers.load_corpus() if clip["spec"]["config"] == "raw"]
    assert raw
    blob = raw[0]["blob"]
    assert runtime.check_clip(blob)[0] == 0
    num_tracks = int(np.frombuffer(blob, dtype=np.uint32, count=1, offset=16)[0])
    # transform_tracks_header (acl_format.h): words 2 .. 7 = animated rotation / translation / scale sub-tracks, constant rotation / translation / scale samples
    for word in range(2, 8):
        for value in (0xFFFFFFFF, 0xFFFFFFFE, 0x80000000, num_tracks + 1):
            status, message = runtime.check_clip(_patched(blob, TRANSFORM_HEADER_OFFSET + 4 * word, "<I", value), check_hash=False)
            assert status != 0, (word, hex(value), message)


def test_database_runtime_headers_must_tile_their_block():
    """clip metadata whose runtime header offsets overlap (or leave a partial segment header) make database_context::initialize write a
    clip hash into another clip's segment header (database.impl.h:151-157); chunk segment headers must name a segment header of the clip
    they name (found by tools/fuzz_gpu_mutated_db.py, round 6)"""
    case = helpers.load_database_golden("three_clips_4k_chunks")
    database, medium, low = case["database"], case["bulk_medium"], case["bulk_low"]
    assert runtime.check_database(database, medium, low)[0] == 0
    words = np.frombuffer(database, dtype=np.uint32)
    num_clips, metadata_offset = int(words[7]), 8 + int(words[9])          # (raw_buffer_header of 8 bytes, then database_header)
    assert num_clips == 3
    for clip in range(num_clips):
        at = metadata_offset + 8 * clip + 4         # database_clip_metadata::clip_header_offset
        original = int(np.frombuffer(database, dtype=np.uint32, count=1, offset=at)[0])
        for delta in (-16, -8, 8, 16, 24):
            status, _ = runtime.check_database(_patched(database, at, "<I", (original + delta) & 0xFFFFFFFF), medium, low, check_hash=False)
            assert status != 0, (clip, delta)
    # a chunk segment header that names the header of ANOTHER segment slot than any of its clip's (first chunk of the medium tier:
    # database_chunk_header {index, size, num_segments} then database_chunk_segment_header {clip_hash, sample_indices, samples_offset, clip_header_offset, segment_header_offset})
    if medium.size:
        at = 12 + 16
        original = int(np.frombuffer(medium, dtype=np.uint32, count=1, offset=at)[0])
        for delta in (-8, 4, 8, 0x10000):
            status, _ = runtime.check_database(database, _patched(medium, at, "<I", (original + delta) & 0xFFFFFFFF), low, check_hash=False)
            assert status != 0, delta


def test_two_chunks_with_keyframes_of_the_same_segment_are_refused():
    """mutation 10915 of seed 72 (tools/fuzz_gpu_mutated_db.py, round 6): a chunk segment header's segment_header_offset moved onto its
    neighbour's -- the later stream-in overwrites the earlier one's tier words in the reference, the patches race on the device"""
    case = helpers.load_database_golden("three_clips_4k_chunks")
    database, medium, low = case["database"], case["bulk_medium"], case["bulk_low"]
    assert medium[4284] == 8
    status, message = runtime.check_database(database, _patched(medium, 4284, "<B", 24), low, check_hash=False)
    assert status != 0 and "same segment" in message, message
