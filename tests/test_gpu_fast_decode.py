"""ACLHIP_DECODE_FAST (aclhip_decompress_params::flags): the opt in tolerance mode of the plain decode. north_star's bar is 1e-5 per
component; the default kernels are bit exact and stay so. With the flag, animated rotations take the hardware's 1 ulp square root /
reciprocal square root and fused multiply-adds: every rotation component within 2e-6 of the oracle, everything else -- translations,
scales, constant and default sub-tracks -- bit identical to it. Asserted over EVERY instance of the BASELINE.json batches (the headline's
64k x 100 bones and the 300-bone rig) and over the real-compressor corpus. Needs a GPU."""
import numpy as np
import pytest
import torch

from acl_amd import runtime
from oracle import bindings as ob
import bench
import helpers

pytestmark = pytest.mark.gpu

ROTATION_TOLERANCE = 2.0e-6
VECTOR_LANES = [4, 5, 6, 8, 9, 10]


def _assert_within_tolerance(fast, expected, what):
    worst = float(np.abs(fast[..., 0:4] - expected[..., 0:4]).max()) if fast.size else 0.0
    assert worst <= ROTATION_TOLERANCE, f"{what}: rotations differ by {worst}"
    assert np.array_equal(np.ascontiguousarray(fast[..., VECTOR_LANES]).view(np.uint32), np.ascontiguousarray(expected[..., VECTOR_LANES]).view(np.uint32)), f"{what}: translations / scales moved"
    return worst


@pytest.mark.parametrize("workload,kernel", [("one_clip", "decompress_tracks_fast_kernel"), ("cinematic", "decompress_tracks_in_turn_fast_kernel")])
def test_every_instance_of_the_baseline_batches(workload, kernel):
    clips, clip_indices, times = bench.build_workload(workload, 0, bench.INSTANCES_PER_GPU)
    bones = clips[0].num_tracks
    with runtime.Context(0) as context:
        handles = np.array([context.register_clip(c.blob) for c in clips], dtype=np.uint32)
        fast_params = runtime.default_params(flags=runtime.DECODE_FAST)
        assert context.tracks_kernel_name(fast_params, pose_stride_bytes=bones * 48) == kernel
        assert not context.tracks_kernel_name(runtime.default_params(), pose_stride_bytes=bones * 48).endswith("fast_kernel")
        d_clips = torch.from_numpy(handles[clip_indices].astype(np.int32)).cuda()
        d_times = torch.from_numpy(times).cuda()
        d_fast = torch.zeros((times.size, bones * 12), dtype=torch.float32, device="cuda")
        d_exact = torch.zeros_like(d_fast)
        stream = torch.cuda.current_stream().cuda_stream
        context.decompress_tracks_batch(d_clips.data_ptr(), d_times.data_ptr(), times.size, d_fast.data_ptr(), bones * 48, params=fast_params, stream=stream)
        context.decompress_tracks_batch(d_clips.data_ptr(), d_times.data_ptr(), times.size, d_exact.data_ptr(), bones * 48, stream=stream)
        torch.cuda.synchronize()
        fast = d_fast.cpu().numpy().reshape(times.size, bones, 12)
        exact = d_exact.cpu().numpy().reshape(times.size, bones, 12)
        # the default kernels == the oracle, bit for bit (a sample of rows here; every row in tests/test_gpu_full_size.py) ...
        rows = np.arange(0, times.size, 257)
        expected = ob.oracle_decompress_tracks_batch([c.blob for c in clips], clip_indices[rows], times[rows], bones)
        assert helpers.bit_equal(exact[rows], expected)
        # ... and the fast ones within the tolerance of them on all 65 536 instances, and of the oracle itself on the sample
        worst = _assert_within_tolerance(fast, exact, workload)
        _assert_within_tolerance(fast[rows], expected, workload)
        assert worst > 0.0      # (the flag does select other arithmetic)
        assert context.rejected_instance_count() == 0


def test_corpus_and_single_track_requests():
    corpus = [clip for clip in helpers.load_corpus() if 0 < clip["spec"]["bones"] <= 104]
    with runtime.Context(0) as context:
        fast_params = runtime.default_params(flags=runtime.DECODE_FAST, rounding_policy=ob.ROUND_NONE)
        for clip in corpus:
            handle = context.register_clip(clip["blob"])
            times, duration = helpers.corpus_sample_times(clip["blob"])
            times = np.minimum(times + np.float32(0.41) / np.float32(clip["spec"]["rate"]), np.float32(duration)).astype(np.float32)
            clips = np.full(times.size, handle, dtype=np.uint32)
            fast = context.decompress_tracks(clips, times, params=fast_params)
            exact = context.decompress_tracks(clips, times)
            _assert_within_tolerance(fast, exact, clip["name"])
            # single track requests: the fast track kernel against the exact whole pose
            bones = clip["spec"]["bones"]
            instance = np.repeat(np.arange(times.size), bones)
            track = np.tile(np.arange(bones, dtype=np.uint32), times.size)
            single = context.decompress_track(clips[instance], times[instance], track, params=runtime.default_params(flags=runtime.DECODE_FAST))
            _assert_within_tolerance(single, exact[instance, track], clip["name"] + " (decompress_track)")
            context.unregister_clip(handle)


def test_the_flag_leaves_other_settings_on_the_exact_kernels():
    """per track rounding, always-normalize, other default modes and output descriptors keep the bit exact kernels: the flag is a hint"""
    clip = next(clip for clip in helpers.load_corpus() if clip["name"].endswith("quant_medium_70x91"))
    with runtime.Context(0) as context:
        handle = context.register_clip(clip["blob"])
        times, _ = helpers.corpus_sample_times(clip["blob"])
        clips = np.full(times.size, handle, dtype=np.uint32)
        for settings, default_mode in ((1, 0), (0, 1), (2, 0)):
            params = helpers.gpu_params(runtime, rounding=ob.ROUND_NONE, settings=settings, default_mode=default_mode)
            exact = context.decompress_tracks(clips, times, params=params, out=np.full((times.size, 70, 12), 3.0, dtype=np.float32))
            params.flags = runtime.DECODE_FAST
            flagged = context.decompress_tracks(clips, times, params=params, out=np.full((times.size, 70, 12), 3.0, dtype=np.float32))
            assert helpers.bit_equal(flagged, exact)
        with pytest.raises(runtime.AclHipError):
            context.decompress_tracks(clips, times, params=runtime.default_params(flags=2))
