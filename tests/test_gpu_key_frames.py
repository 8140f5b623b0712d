"""The seek at every key frame of a clip and on both sides of it (k / rate, its float neighbours, the middle of every interval, both ends
from outside) against the oracle, for clips of many lengths -- both of the compressor's cuts (compression/impl/segment_streams.h: the
samples of a short last segment spread over the first ones, or kept), other segment sizes of the same splitter, and cuts the
compressor never makes (a segment start moved by hand: the sample records follow the clip's own segment_start_indices) -- through the
pose kernels of every layout, the kernel of poses of several windows, every rounding policy and single bone requests. Needs a GPU."""
import numpy as np
import pytest
import torch

from acl_amd import runtime, synth
from oracle import bindings as ob
import helpers

pytestmark = pytest.mark.gpu

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


def times_around_every_key_frame(clip, wrap_extra=0):
    """k / rate, and its float neighbours, for every sample index of the clip (and one beyond for clips that wrap); the ends, outside"""
    rate = np.float32(clip.sample_rate)
    at = (np.arange(clip.num_samples + wrap_extra, dtype=np.float32) / rate).astype(np.float32)
    below = np.nextafter(at, np.float32(-np.inf)).astype(np.float32)
    above = np.nextafter(at, np.float32(np.inf)).astype(np.float32)
    middle = (at + np.float32(0.5) / rate).astype(np.float32)
    return np.concatenate([at, below, above, middle, np.array([-1.0, 1.0e9], dtype=np.float32)])


def decode_compact(context, handle, times, num_tracks, layout):
    bytes_per_track = runtime.LAYOUTS[layout][1]
    n = times.size
    stride = (num_tracks * bytes_per_track + 15) // 16 * 16         # rows are 16 byte aligned
    d_clips = torch.full((n,), int(handle), dtype=torch.int32, device="cuda")
    d_times = torch.from_numpy(times).cuda()
    poses = torch.zeros((n, stride // 4), dtype=torch.float32, device="cuda")
    output = runtime.OutputDesc()
    output.layout = runtime.LAYOUTS[layout][0]
    context.decompress_tracks_batch_out(d_clips.data_ptr(), d_times.data_ptr(), n, poses.data_ptr(), stride, output)
    torch.cuda.synchronize()
    return poses.cpu().numpy()[:, : num_tracks * bytes_per_track // 4].reshape(n, num_tracks, bytes_per_track // 4)


def check_clip_everywhere(context, clip, blob=None, check_hash=True, layouts=("qvv40", "qv32")):
    blob = clip.blob if blob is None else blob
    handle = context.register_clip(blob, check_hash=check_hash)
    wraps = bool(clip.spec.wrap) and clip.spec.version > 7
    times = times_around_every_key_frame(clip, 2 if wraps else 0)
    n = times.size
    expected = ob.oracle_decompress_tracks_batch([blob], np.zeros(n, dtype=np.uint32), times, clip.num_tracks)
    got = context.decompress_tracks(np.full(n, handle, dtype=np.uint32), times)
    assert helpers.exact(got, expected)
    for rounding in (1, 2, 3):
        got = context.decompress_tracks(np.full(n, handle, dtype=np.uint32), times, params=runtime.default_params(rounding_policy=rounding))
        assert helpers.exact(got, ob.oracle_decompress_tracks_batch([blob], np.zeros(n, dtype=np.uint32), times, clip.num_tracks, rounding=rounding)), rounding
    for layout in layouts:
        got = decode_compact(context, handle, times, clip.num_tracks, layout)
        assert helpers.exact(got, runtime.relayout_pose(expected, runtime.LAYOUTS[layout][0])), layout
    # single bone requests share the seek
    tracks = (np.arange(n) % max(clip.num_tracks, 1)).astype(np.uint32)
    single = context.decompress_track(np.full(n, handle, dtype=np.uint32), times, tracks)
    assert helpers.exact(single, expected[np.arange(n), tracks])
    context.unregister_clip(handle)


@pytest.mark.parametrize("num_samples", [17, 31, 32, 33, 47, 48, 49, 64, 65, 100, 161, 301, 1000])
def test_every_key_frame_of_clips_of_many_lengths(num_samples):
    with runtime.Context(0) as context:
        clip = synth.build_clip(seed=40 + num_samples, num_tracks=37, num_samples=num_samples, has_scale=num_samples % 2, wrap=int(num_samples % 3 == 0), strip_keyframes=int(num_samples % 5 == 0))
        check_clip_everywhere(context, clip)
        assert context.rejected_instance_count() == 0


@pytest.mark.parametrize("ideal,maximum,num_samples", [(8, 15, 203), (4, 7, 99), (20, 32, 211), (31, 32, 100), (2, 3, 41), (16, 16, 97)])
def test_other_segment_sizes(ideal, maximum, num_samples):
    with runtime.Context(0) as context:
        clip = synth.build_clip(seed=7, num_tracks=21, num_samples=num_samples, ideal_segment_samples=ideal, max_segment_samples=maximum)
        check_clip_everywhere(context, clip)


def test_poses_of_several_windows():
    """the 300-bone rig's kernel (items in turn, 16 byte key reads)"""
    with runtime.Context(0) as context:
        clip = synth.build_clip(seed=77, num_tracks=300, num_samples=100, has_scale=1, scale_default=0.4)
        check_clip_everywhere(context, clip)


def test_cuts_the_compressor_never_makes():
    with runtime.Context(0) as context:
        clip = synth.build_clip(seed=5, num_tracks=24, num_samples=100)          # 17 17 17 17 16 16
        for segment, delta in ((1, -1), (2, 1), (4, -1)):
            moved = with_moved_start(clip, segment, delta)
            check_clip_everywhere(context, clip, blob=moved, check_hash=False)
        # both kinds in one batch
        regular = context.register_clip(clip.blob)
        moved = with_moved_start(clip, 2, 1)
        other = context.register_clip(moved, check_hash=False)
        rng = np.random.default_rng(3)
        which = rng.integers(0, 2, size=2000)
        times = rng.uniform(-0.1, clip.duration + 0.1, size=2000).astype(np.float32)
        got = context.decompress_tracks(np.where(which == 0, regular, other).astype(np.uint32), times)
        expected = ob.oracle_decompress_tracks_batch([clip.blob, moved], which.astype(np.uint32), times, clip.num_tracks)
        assert helpers.exact(got, expected)
        assert context.rejected_instance_count() == 0
