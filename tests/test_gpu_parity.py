"""Parity of the HIP path (through the C ABI) with the CPU oracle and the committed reference golden vectors. Needs a GPU."""
import ctypes
import os

import numpy as np
import pytest

from acl_amd import runtime, synth
from oracle import bindings as ob
import helpers
from conftest import CLIP_SPECS, sample_times_for

pytestmark = pytest.mark.gpu

TOLERANCE = 1.0e-5      # BASELINE.json: within 1e-5 per component of the reference CPU decompress_tracks()


@pytest.fixture(scope="module", params=["common_case_kernel", "any_settings_kernel"])
def context(request):
    """Every test runs twice: with the launch free to pick the common case kernel, and pinned to the any-settings kernel."""
    if request.param == "any_settings_kernel":
        os.environ["ACLHIP_FORCE_GENERIC_KERNEL"] = "1"
    try:
        ctx = runtime.Context(0)
    finally:
        os.environ.pop("ACLHIP_FORCE_GENERIC_KERNEL", None)
    yield ctx
    ctx.close()


def test_native_library_is_the_in_tree_hip_build():
    assert os.path.samefile(runtime.library_path(), os.path.join(os.path.dirname(runtime.__file__), "lib", "libaclhip.so"))
    assert runtime.load_library() is not None


@pytest.mark.parametrize("name", sorted(CLIP_SPECS))
def test_decompress_tracks_matches_oracle(context, name):
    clip = synth.build_clip(**CLIP_SPECS[name])
    handle = context.register_clip(clip.blob)
    info = context.clip_info(handle)
    assert info.num_tracks == clip.num_tracks
    assert info.duration == ob.oracle().aclo_finite_duration(clip.blob.ctypes.data, ob.LOOP_AS_COMPRESSED)
    rng = np.random.default_rng(11)
    times = sample_times_for(info.duration, 200, rng)
    for policy in (0, 1, 2, 3):
        poses = context.decompress_tracks(np.full(times.size, handle), times, params=runtime.default_params(rounding_policy=policy))
        exact = 0
        for i, t in enumerate(times):
            expected = ob.oracle_decompress_tracks(clip.blob, float(t), policy)
            assert helpers.max_abs_diff(poses[i], expected) <= TOLERANCE, f"{name} policy {policy} t {t}"
            exact += int(np.array_equal(poses[i].view(np.uint32), expected.view(np.uint32)))
        # the kernel follows the reference operation order with contraction off: in practice every pose is bit identical
        assert exact == times.size, f"{name} policy {policy}: only {exact}/{times.size} poses bit exact"
    context.unregister_clip(handle)
    assert context.rejected_instance_count() == 0


@pytest.mark.parametrize("name", sorted(CLIP_SPECS))
def test_decompress_track_matches_oracle_and_whole_pose(context, name):
    clip = synth.build_clip(**CLIP_SPECS[name])
    handle = context.register_clip(clip.blob)
    info = context.clip_info(handle)
    rng = np.random.default_rng(12)
    times = sample_times_for(info.duration, 120, rng)
    tracks = rng.integers(0, info.num_tracks, size=times.size).astype(np.uint32)
    single = context.decompress_track(np.full(times.size, handle), times, tracks)
    poses = context.decompress_tracks(np.full(times.size, handle), times)
    for i, t in enumerate(times):
        expected = ob.oracle_decompress_track(clip.blob, float(t), int(tracks[i]))
        assert helpers.max_abs_diff(single[i], expected) <= TOLERANCE
        # validate_tracks.cpp:231-258: decompress_track == decompress_tracks (vec3 exactly; here the quaternion as well)
        assert np.array_equal(single[i].view(np.uint32), poses[i, tracks[i]].view(np.uint32))
    context.unregister_clip(handle)


@pytest.mark.parametrize("name", helpers.golden_cases())
def test_matches_reference_golden_vectors(context, name):
    """Golden poses come from the reference's own headers (tests/golden/make_golden.py)."""
    case = helpers.load_golden(name)
    handle = context.register_clip(case["blob"])
    times = case["times"]
    n = times.size
    for p, policy in enumerate(case["policies"]):
        params = helpers.gpu_params(runtime, int(policy), case["settings"], case["default_mode"])
        out = np.repeat(case["prefill"][None], n, axis=0).copy()
        poses = context.decompress_tracks(np.full(n, handle), times, params=params, out=out,
                                          default_values=case["defaults"] if case["default_mode"] in (2, 3) else None,
                                          track_rounding=case["track_rounding"])
        assert helpers.max_abs_diff(poses, case["poses"][p]) <= TOLERANCE, f"{name} policy {policy}"
        assert helpers.bit_equal(poses, case["poses"][p]), f"{name} policy {policy}: not bit exact vs the reference"

        params = helpers.gpu_params(runtime, int(policy), case["settings"], case["default_mode"])
        single_out = case["prefill"][case["track_indices"]].copy()
        single = context.decompress_track(np.full(n, handle), times, case["track_indices"], params=params, out=single_out,
                                          default_values=case["defaults"] if case["default_mode"] in (2, 3) else None,
                                          track_rounding=case["track_rounding"])
        assert helpers.max_abs_diff(single, case["single"][p]) <= TOLERANCE, f"{name} policy {policy} (single track)"
    context.unregister_clip(handle)


@pytest.mark.parametrize("settings", [1, 2])
def test_normalization_and_per_track_rounding(context, settings):
    clip = synth.build_clip(**CLIP_SPECS["raw_and_constant_rates"])
    handle = context.register_clip(clip.blob)
    info = context.clip_info(handle)
    rng = np.random.default_rng(13)
    track_rounding = rng.integers(0, 4, size=info.num_tracks).astype(np.uint8)
    times = sample_times_for(info.duration, 100, rng)
    for policy in (0, 1, 2, 3, 4):
        params = helpers.gpu_params(runtime, policy, settings)
        poses = context.decompress_tracks(np.full(times.size, handle), times, params=params, track_rounding=track_rounding)
        options = helpers.oracle_options(settings, 0, None, track_rounding)
        for i, t in enumerate(times):
            expected = ob.oracle_decompress_tracks(clip.blob, float(t), policy, options)
            assert np.array_equal(poses[i].view(np.uint32), expected.view(np.uint32)), f"settings {settings} policy {policy}"
    context.unregister_clip(handle)


def test_per_instance_rounding_policies(context):
    clip = synth.build_clip(**CLIP_SPECS["cmu_70_default"])
    handle = context.register_clip(clip.blob)
    info = context.clip_info(handle)
    rng = np.random.default_rng(14)
    times = sample_times_for(info.duration, 200, rng)
    policies = rng.integers(0, 4, size=times.size).astype(np.uint8)
    poses = context.decompress_tracks(np.full(times.size, handle), times, instance_rounding=policies)
    for i, t in enumerate(times):
        expected = ob.oracle_decompress_tracks(clip.blob, float(t), int(policies[i]))
        assert np.array_equal(poses[i].view(np.uint32), expected.view(np.uint32))
    context.unregister_clip(handle)


@pytest.mark.parametrize("looping", [runtime.LOOP_CLAMP, runtime.LOOP_WRAP])
def test_looping_policy_override(context, looping):
    clip = synth.build_clip(**CLIP_SPECS["wrap_77"])
    handle = context.register_clip(clip.blob)
    duration = ob.oracle().aclo_finite_duration(clip.blob.ctypes.data, looping)
    rng = np.random.default_rng(15)
    times = sample_times_for(duration, 100, rng)
    poses = context.decompress_tracks(np.full(times.size, handle), times, params=runtime.default_params(looping_policy=looping))
    options = helpers.oracle_options(looping=looping)
    for i, t in enumerate(times):
        expected = ob.oracle_decompress_tracks(clip.blob, float(t), 0, options)
        assert np.array_equal(poses[i].view(np.uint32), expected.view(np.uint32))
    context.unregister_clip(handle)


def test_mixed_clips_in_one_batch(context):
    names = ["cmu_100", "scale_37", "stripped_wrap_scale", "one_sample", "all_default", "cinematic_300"]
    clips = [synth.build_clip(**CLIP_SPECS[n]) for n in names]
    handles = [context.register_clip(c.blob) for c in clips]
    rng = np.random.default_rng(16)
    n = 500
    which = rng.integers(0, len(clips), size=n)
    durations = np.array([context.clip_info(h).duration for h in handles], dtype=np.float32)
    times = (rng.uniform(0, 1, size=n).astype(np.float32) * durations[which]).astype(np.float32)
    max_tracks = max(c.num_tracks for c in clips)
    out = np.full((n, max_tracks, 12), -7.0, dtype=np.float32)
    poses = context.decompress_tracks(np.array(handles, dtype=np.uint32)[which], times, num_tracks=max_tracks, out=out)
    for i in range(n):
        clip = clips[which[i]]
        expected = ob.oracle_decompress_tracks(clip.blob, float(times[i]))
        assert np.array_equal(poses[i, :clip.num_tracks].view(np.uint32), expected.view(np.uint32))
        assert np.all(poses[i, clip.num_tracks:] == -7.0)        # a smaller clip never writes past its own tracks
    for h in handles:
        context.unregister_clip(h)


def test_invalid_inputs_are_rejected(context):
    clip = synth.build_clip(num_tracks=8, num_samples=12)
    # corrupted blob: hash check (compressed_tracks::is_valid(true))
    bad = synth.aligned_bytes(clip.blob.size)
    bad[:] = clip.blob
    bad[200] ^= 0xFF
    with pytest.raises(runtime.AclHipError) as excinfo:
        context.register_clip(bad)
    assert excinfo.value.status == 2
    handle = context.register_clip(bad, check_hash=False)      # initialize() does not check the hash by default
    context.unregister_clip(handle)
    truncated = synth.aligned_bytes(40)
    truncated[:] = clip.blob[:40]
    with pytest.raises(runtime.AclHipError):
        context.register_clip(truncated)

    handle = context.register_clip(clip.blob)
    assert context.clip_matches(handle, clip.blob)
    before = context.rejected_instance_count()
    # unknown clip handle and out-of-range track index: the instance is skipped like the reference's silent returns
    out = np.full((2, 8, 12), 3.0, dtype=np.float32)
    context.decompress_tracks(np.array([handle, 12345], dtype=np.uint32), np.zeros(2, dtype=np.float32), num_tracks=8, out=out)
    assert np.all(out[1] == 3.0) and not np.all(out[0] == 3.0)
    single = np.full((1, 12), 3.0, dtype=np.float32)
    context.decompress_track(np.array([handle], dtype=np.uint32), np.zeros(1, dtype=np.float32), np.array([99], dtype=np.uint32), out=single)
    assert np.all(single == 3.0)
    assert context.rejected_instance_count() == before + 2
    with pytest.raises(runtime.AclHipError):
        context.decompress_tracks(np.array([handle], dtype=np.uint32), np.zeros(1, dtype=np.float32), params=runtime.default_params(rounding_policy=runtime.ROUND_PER_TRACK))
    context.unregister_clip(handle)
    with pytest.raises(runtime.AclHipError):
        context.unregister_clip(handle)


def test_empty_batch_and_empty_clip(context):
    context.decompress_tracks(np.zeros(0, dtype=np.uint32), np.zeros(0, dtype=np.float32), num_tracks=0)
    clip = synth.build_clip(num_tracks=0, num_samples=0)
    handle = context.register_clip(clip.blob)
    out = np.full((1, 4, 12), 5.0, dtype=np.float32)
    context.decompress_tracks(np.array([handle], dtype=np.uint32), np.zeros(1, dtype=np.float32), num_tracks=4, out=out)
    assert np.all(out == 5.0)       # empty track list: nothing is written (decompression.transform.h:1531-1533)
    context.unregister_clip(handle)


def test_non_finite_sample_times_clamp_like_the_reference(context):
    """scalar_clamp(sample_time, 0, duration) (decompression.transform.h:215-216) with NaN / infinite / huge times: the oracle agrees
    with the reference on these (tests/test_oracle_vs_reference.py), the kernels must agree with the oracle."""
    clip = synth.build_clip(**CLIP_SPECS["cmu_70_default"])
    handle = context.register_clip(clip.blob)
    times = np.array([np.nan, np.inf, -np.inf, 1e30, -1e30, -0.0], dtype=np.float32)
    for policy in (0, 1, 2, 3):
        poses = context.decompress_tracks(np.full(times.size, handle), times, params=runtime.default_params(rounding_policy=policy))
        for i, t in enumerate(times):
            expected = ob.oracle_decompress_tracks(clip.blob, float(t), policy)
            assert helpers.bit_equal(poses[i], expected), f"time {t} policy {policy}"
    context.unregister_clip(handle)


def test_context_is_usable_from_several_host_threads(context):
    """Registration, host-pointer decodes and unregistration from four threads at once on one context (ctypes releases the GIL):
    every thread must get its own clips' poses, bit for bit."""
    import threading
    errors = []

    def worker(thread_index):
        try:
            rng = np.random.default_rng(1000 + thread_index)
            for iteration in range(12):
                clip = synth.build_clip(seed=5000 + thread_index * 100 + iteration, num_tracks=int(rng.integers(3, 40)), num_samples=int(rng.integers(2, 70)))
                handle = context.register_clip(clip.blob)
                times = rng.uniform(0.0, clip.duration, size=16).astype(np.float32)
                poses = context.decompress_tracks(np.full(16, handle, dtype=np.uint32), times)
                for i in range(16):
                    if not helpers.bit_equal(poses[i], ob.oracle_decompress_tracks(clip.blob, float(times[i]))):
                        errors.append((thread_index, iteration, i))
                context.unregister_clip(handle)
        except Exception as error:      # noqa: BLE001 -- reported below
            errors.append((thread_index, repr(error)))

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(4)]
    for thread in threads:
        thread.start()
    for thread in threads:
        thread.join(timeout=120)
    assert not any(thread.is_alive() for thread in threads)
    assert errors == []


@pytest.mark.parametrize("num_requests", [1, 63, 64, 65, 1000, 4097])
def test_track_requests_with_refused_and_skipped_lanes_inside_full_waves(context, num_requests):
    """decompress_track_kernel works on 64 requests per wave (round 5): a wave that holds a refused request -- an unknown or scalar clip,
    a track index past the clip's end -- or a request whose default sub-tracks are skipped leaves those bytes as the caller had them and
    still decodes its other lanes to the oracle's bits; waves of one clip and waves of mixed clips; batch sizes around the wave size."""
    rng = np.random.default_rng(num_requests)
    clips = [synth.build_clip(**CLIP_SPECS["cmu_100"]), synth.build_clip(**CLIP_SPECS["scale_37"]), synth.build_clip(**CLIP_SPECS["all_default"])]
    curves = synth.build_scalar_clip(seed=3, track_type=0, num_tracks=10, num_samples=20)
    handles = [context.register_clip(c.blob) for c in clips]
    scalar_handle = context.register_clip(curves.blob)
    for mixed in (False, True):
        which = rng.integers(0, len(clips), size=num_requests) if mixed else np.zeros(num_requests, dtype=np.int64)
        clip_ids = np.array([handles[w] for w in which], dtype=np.uint32)
        times = np.array([rng.uniform(0.0, clips[w].duration) for w in which], dtype=np.float32)
        tracks = np.array([rng.integers(0, clips[w].num_tracks) for w in which], dtype=np.uint32)
        # a tenth of the requests is refused, one way or another
        refused = rng.uniform(size=num_requests) < 0.1
        kinds = rng.integers(0, 3, size=num_requests)
        clip_ids[refused & (kinds == 0)] = 54321
        clip_ids[refused & (kinds == 1)] = scalar_handle
        tracks[refused & (kinds == 2)] = 5000
        for default_mode in (ob.DEFAULT_CONSTANT, ob.DEFAULT_SKIPPED):
            modes = dict(default_rotation_mode=default_mode, default_translation_mode=default_mode, default_scale_mode=default_mode if default_mode == ob.DEFAULT_SKIPPED else ob.DEFAULT_LEGACY)
            params = runtime.default_params(**modes)
            options = ob.default_options(**modes)
            before = context.rejected_instance_count()
            fill = np.float32(-7.5)
            got = context.decompress_track(clip_ids, times, tracks, params=params, out=np.full((num_requests, 12), fill, dtype=np.float32))
            expected = np.full((num_requests, 12), fill, dtype=np.float32)
            for i in range(num_requests):
                if not refused[i]:
                    pose = ob.oracle_decompress_tracks(clips[which[i]].blob, float(times[i]), options=options, out=np.full((clips[which[i]].num_tracks, 12), fill, dtype=np.float32))
                    expected[i] = pose[tracks[i]]
            assert helpers.bit_equal(got, expected), (num_requests, mixed, default_mode)
            assert context.rejected_instance_count() == before + int(refused.sum())
    for handle in handles + [scalar_handle]:
        context.unregister_clip(handle)


@pytest.mark.parametrize("num_requests", [300, 5000])
def test_track_requests_in_the_librarys_locality_order_decode_to_the_same_transforms(context, num_requests):
    """aclhip_order_track_requests_for_locality permutes a request list (clips bucketed, every clip on one XCD): request k of the ordered
    launch is request order[k] of the caller's list -- the same bits, wherever a request sits in a wave"""
    rng = np.random.default_rng(num_requests)
    clips = [synth.build_clip(**CLIP_SPECS[name]) for name in ("cmu_100", "scale_37", "stripped_wrap_scale", "two_segments_32", "cmu_70_default")]
    handles = np.array([context.register_clip(c.blob) for c in clips], dtype=np.uint32)
    which = rng.integers(0, len(clips), size=num_requests)
    times = np.array([rng.uniform(0.0, clips[w].duration) for w in which], dtype=np.float32)
    tracks = np.array([rng.integers(0, clips[w].num_tracks) for w in which], dtype=np.uint32)
    as_drawn = context.decompress_track(handles[which], times, tracks)
    order = runtime.order_track_requests_for_locality(handles[which])
    assert np.array_equal(np.sort(order), np.arange(num_requests))
    ordered = context.decompress_track(handles[which][order], times[order], tracks[order])
    assert helpers.bit_equal(ordered, as_drawn[order])
    for i in rng.integers(0, num_requests, size=40):
        assert helpers.bit_equal(as_drawn[i], ob.oracle_decompress_track(clips[which[i]].blob, float(times[i]), int(tracks[i])))
    for handle in handles:
        context.unregister_clip(int(handle))
