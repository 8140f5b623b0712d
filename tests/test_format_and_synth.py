"""Host logic: the restated binary layout and the synthetic clip writer (no GPU)."""
import ctypes

import numpy as np
import pytest

from acl_amd import synth
from oracle import bindings as ob
from conftest import CLIP_SPECS


@pytest.mark.parametrize("name", sorted(CLIP_SPECS))
def test_synthetic_clip_is_a_valid_compressed_tracks(name):
    clip = synth.build_clip(with_side_data=True, **CLIP_SPECS[name])
    blob = clip.blob
    assert blob.ctypes.data % 16 == 0
    assert ob.oracle().aclo_is_valid(blob.ctypes.data, blob.size, 1) == 0      # tag, version, FNV-1a hash
    size, stored_hash = np.frombuffer(blob[:8].tobytes(), dtype=np.uint32)
    assert size == blob.size
    assert stored_hash == ob.oracle().aclo_hash32(blob[8:].ctypes.data, blob.size - 8)
    assert bytes(blob[-15:]) == b"\0" * 15                                    # 15 bytes of padding (compress.transform.impl.h:396)
    assert ob.oracle().aclo_num_tracks(blob.ctypes.data) == clip.num_tracks
    assert ob.oracle().aclo_finite_duration(blob.ctypes.data, ob.LOOP_AS_COMPRESSED) == pytest.approx(clip.duration, abs=0)


def test_corrupting_a_byte_breaks_the_hash():
    clip = synth.build_clip(num_tracks=10, num_samples=10)
    blob = clip.blob.copy()
    blob = synth.aligned_bytes(blob.size)
    blob[:] = clip.blob
    blob[100] ^= 0x40
    assert ob.oracle().aclo_is_valid(blob.ctypes.data, blob.size, 1) == 8
    assert ob.oracle().aclo_is_valid(blob.ctypes.data, blob.size, 0) == 0


def test_invalid_spec_is_rejected():
    with pytest.raises(ValueError):
        synth.build_clip(version=3)
    with pytest.raises(ValueError):
        synth.build_clip(num_samples=0)


def test_generation_is_deterministic():
    a = synth.build_clip(seed=11, num_tracks=40)
    b = synth.build_clip(seed=11, num_tracks=40)
    c = synth.build_clip(seed=12, num_tracks=40)
    assert np.array_equal(a.blob, b.blob)
    assert not np.array_equal(a.blob[: min(a.blob.size, c.blob.size)], c.blob[: min(a.blob.size, c.blob.size)])


@pytest.mark.parametrize("name", ["cmu_100", "single_segment", "scale_37", "stripped", "wrap_77", "raw_and_constant_rates", "high_bits_23"])
def test_decoding_at_stored_keyframes_reproduces_the_quantized_samples(name):
    """Independent check of writer + decoder: at a stored keyframe the decode must equal the dequantized value the writer
    computed in double precision, without any interpolation."""
    clip = synth.build_clip(with_side_data=True, **CLIP_SPECS[name])
    for k in range(clip.num_samples):
        if not clip.stored_keyframes[k]:
            continue
        t = float(np.float32(k) / np.float32(clip.sample_rate))
        pose = ob.oracle_decompress_tracks(clip.blob, t, ob.ROUND_NEAREST)
        expected = clip.expected_keyframes[k]
        # xyz of everything; quaternion w is sqrt(1 - |xyz|^2): ill conditioned near 0, bounded separately
        assert np.abs(pose[:, [0, 1, 2, 4, 5, 6, 8, 9, 10]] - expected[:, [0, 1, 2, 4, 5, 6, 8, 9, 10]]).max() <= 2e-6
        assert np.abs(pose[:, 3] - expected[:, 3]).max() <= 1e-3
        assert np.abs(np.linalg.norm(pose[:, :4], axis=1) - 1.0).max() <= 1e-5


def test_lossy_error_is_bounded_by_the_quantization_step():
    clip = synth.build_clip(with_side_data=True, min_bits=14, max_bits=16, width0_fraction=0.0, raw_fraction=0.0, num_samples=60)
    worst = 0.0
    for k in range(clip.num_samples):
        t = float(np.float32(k) / np.float32(clip.sample_rate))
        pose = ob.oracle_decompress_tracks(clip.blob, t, ob.ROUND_NEAREST)
        worst = max(worst, float(np.abs(pose[:, [0, 1, 2, 4, 5, 6]] - clip.raw_keyframes[k][:, [0, 1, 2, 4, 5, 6]]).max()))
    # >= 14 bits over a range of at most 4 units, two range reductions: comfortably below 1e-3
    assert worst < 1e-3


def test_segment_layout_follows_the_reference_split_rule():
    # compression/impl/segment.transform.h:65-128: <= 31 samples is one segment; 301 samples -> 19 estimated segments, the 13
    # samples of the last one are spread over the others -> 18 segments of 16/17
    def num_segments(blob):
        return int(np.frombuffer(blob[32:36].tobytes(), dtype=np.uint32)[0])
    assert num_segments(synth.build_clip(num_samples=31, num_tracks=4).blob) == 1
    assert num_segments(synth.build_clip(num_samples=32, num_tracks=4).blob) == 2
    clip = synth.build_clip(num_samples=301, num_tracks=4)
    assert num_segments(clip.blob) == 18
    starts = np.frombuffer(clip.blob[32 + 52: 32 + 52 + 4 * 19].tobytes(), dtype=np.uint32)
    assert starts[0] == 0 and starts[-1] == 0xFFFFFFFF
    sizes = np.diff(np.concatenate([starts[:-1], [301]]).astype(np.int64))
    assert sizes.min() >= 16 and sizes.max() <= 17 and sizes.sum() == 301
