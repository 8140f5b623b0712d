"""Registration's segment analysis (host only: aclhip_analyze_clip): clips cut the way the reference's compressor cuts
(compression/impl/segment_streams.h, split_samples_per_segment: segments of 16, the samples of a short last segment spread over the
first ones or kept) have the REGULAR_SEGMENTS fact -- the kernels then find a key's segment by arithmetic (aclhip_device.h,
segment_of_key_frame) --, any other cut does not and keeps taking the segment from the sample records."""
import numpy as np
import pytest

from acl_amd import runtime, synth

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


@pytest.mark.parametrize("num_samples", list(range(1, 70)) + [95, 96, 97, 100, 127, 128, 129, 301, 1000, 4097])
def test_compressor_cuts_are_regular(num_samples):
    clip = synth.build_clip(seed=num_samples, num_tracks=5, num_samples=num_samples)
    assert runtime.analyze_clip(clip.blob) & runtime.CLIP_FACT_REGULAR_SEGMENTS


@pytest.mark.parametrize("ideal,maximum", [(8, 15), (16, 16), (4, 7), (20, 32), (31, 32), (2, 3)])
def test_other_segment_sizes_of_the_same_splitter_are_regular(ideal, maximum):
    for num_samples in (ideal, ideal + 1, 3 * ideal - 1, 3 * ideal, 3 * ideal + 1, 10 * ideal + ideal // 2, 333):
        clip = synth.build_clip(seed=3, num_tracks=4, num_samples=num_samples, ideal_segment_samples=ideal, max_segment_samples=maximum)
        assert runtime.analyze_clip(clip.blob) & runtime.CLIP_FACT_REGULAR_SEGMENTS, (ideal, maximum, num_samples)


def test_a_cut_the_compressor_never_makes_is_not():
    clip = synth.build_clip(seed=5, num_tracks=6, num_samples=100)          # 17 17 17 17 16 16
    starts = segment_starts(clip.blob.copy())
    sizes = np.diff(np.append(starts, 100))
    assert sizes.tolist() == [17, 17, 17, 17, 16, 16]
    for segment, delta in ((1, -1), (2, 1), (4, -1), (5, -2)):
        moved = with_moved_start(clip, segment, delta)
        status, message = runtime.check_clip(moved, check_hash=False)
        assert status == 0, message
        assert not runtime.analyze_clip(moved, check_hash=False) & runtime.CLIP_FACT_REGULAR_SEGMENTS, (segment, delta)
    # ... while a short LAST segment is the compressor's other form: 16 16 16 16 16 16 4 -> 16 16 16 16 16 17 3 is not, 17 x 5 + 15 is
    moved = with_moved_start(clip, 5, 1)        # 17 17 17 17 17 15: the first five hold 17, the last at most that
    assert runtime.analyze_clip(moved, check_hash=False) & runtime.CLIP_FACT_REGULAR_SEGMENTS


def test_the_arithmetic_matches_the_table_on_every_sample():
    """segment_of_key_frame (aclhip_device.h) restated: mulhi(n, ceil(2^32 / A)) below R A, mulhi(n - R (A - B), ceil(2^32 / B)) from there on,
    against the start indices of the clip, for every sample of clips of many lengths"""
    for num_samples in list(range(33, 400, 7)) + [5000]:
        clip = synth.build_clip(seed=1, num_tracks=2, num_samples=num_samples)
        starts = segment_starts(clip.blob.copy()).astype(np.int64)
        sizes = np.diff(np.append(starts, num_samples))
        leading_size, leading = int(sizes[0]), 0
        while leading + 1 < len(sizes) and sizes[leading] == leading_size:
            leading += 1
        trailing_size = max(int(sizes[leading]), leading_size) if leading + 1 == len(sizes) else int(sizes[leading])
        magic = [-(-(1 << 32) // leading_size), -(-(1 << 32) // trailing_size)]
        split, shift = leading * leading_size, leading * (leading_size - trailing_size)
        samples = np.arange(num_samples, dtype=np.int64)
        computed = np.where(samples < split, (samples * magic[0]) >> 32, ((samples - shift) * magic[1]) >> 32)
        expected = np.searchsorted(starts, samples, side="right") - 1
        assert np.array_equal(computed, expected), num_samples
