"""Scalar track lists (float1f / float2f / float3f / float4f / vector4f) through the C ABI: parity with the CPU oracle and with the
reference's golden vectors. Needs a GPU."""
import numpy as np
import pytest

from acl_amd import runtime, synth
from oracle import bindings as ob
import helpers

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def context():
    ctx = runtime.Context(0)
    yield ctx
    ctx.close()


@pytest.mark.parametrize("name", sorted(helpers.SCALAR_CLIP_SPECS))
def test_decompress_tracks_and_track_match_oracle(context, name):
    clip = synth.build_scalar_clip(**helpers.SCALAR_CLIP_SPECS[name])
    handle = context.register_clip(clip.blob)
    info = context.clip_info(handle)
    assert (info.num_tracks, info.num_samples, info.track_type, info.num_components) == (clip.num_tracks, clip.num_samples, clip.track_type, clip.num_components)
    assert info.duration == np.float32(clip.duration)
    rng = np.random.default_rng(12)
    times = np.concatenate([rng.uniform(-0.1, clip.duration + 0.1, size=120), [0.0, clip.duration]]).astype(np.float32)
    handles = np.full(times.size, handle, dtype=np.uint32)
    for policy in (0, 1, 2, 3):
        values = context.decompress_scalar_tracks(handles, times, params=runtime.default_params(rounding_policy=policy))
        for i, t in enumerate(times):
            assert helpers.exact(values[i], ob.oracle_scalar_decompress_tracks(clip.blob, float(t), policy)), f"{name} policy {policy} t {t}"
    tracks = rng.integers(0, clip.num_tracks, size=times.size).astype(np.uint32)
    single = context.decompress_scalar_track(handles, times, tracks)
    whole = context.decompress_scalar_tracks(handles, times)
    for i in range(times.size):
        assert helpers.exact(single[i], whole[i, tracks[i]])
    for looping in (runtime.LOOP_CLAMP, runtime.LOOP_WRAP):
        values = context.decompress_scalar_tracks(handles, times, params=runtime.default_params(looping_policy=looping))
        for i in range(0, times.size, 5):
            assert helpers.exact(values[i], ob.oracle_scalar_decompress_tracks(clip.blob, float(times[i]), 0, ob.default_options(looping_policy=looping)))
    context.unregister_clip(handle)
    assert context.rejected_instance_count() == 0


@pytest.mark.parametrize("name", helpers.scalar_golden_cases())
def test_matches_reference_golden_vectors(context, name):
    case = helpers.load_scalar_golden(name)
    handle = context.register_clip(case["blob"])
    times = case["times"]
    handles = np.full(times.size, handle, dtype=np.uint32)
    for p, policy in enumerate(case["policies"]):
        params = runtime.default_params(rounding_policy=int(policy), per_track_rounding=1 if policy == runtime.ROUND_PER_TRACK else 0)
        track_rounding = case["track_rounding"] if policy == runtime.ROUND_PER_TRACK else None
        values = context.decompress_scalar_tracks(handles, times, params=params, track_rounding=track_rounding)
        assert helpers.exact(values, case["values"][p]), f"{name}: policy {policy}"
        params = runtime.default_params(rounding_policy=int(policy), per_track_rounding=1 if policy == runtime.ROUND_PER_TRACK else 0)
        single = context.decompress_scalar_track(handles, times, case["track_indices"], params=params, track_rounding=track_rounding)
        assert helpers.exact(single, case["single"][p])
    for looping, key in ((runtime.LOOP_CLAMP, "values_clamp"), (runtime.LOOP_WRAP, "values_wrap")):
        values = context.decompress_scalar_tracks(handles, times, params=runtime.default_params(looping_policy=looping))
        assert helpers.exact(values, case[key])
    context.unregister_clip(handle)


def test_mixed_track_types_and_per_instance_rounding_in_one_batch(context):
    clips = [synth.build_scalar_clip(**helpers.SCALAR_CLIP_SPECS[name]) for name in ("float1f_all_rates", "float3f_wrap", "vector4f_low_bits", "float2f_v2_0_rates")]
    handles = [context.register_clip(clip.blob) for clip in clips]
    rng = np.random.default_rng(13)
    n = 400
    which = rng.integers(0, len(clips), size=n)
    times = np.array([rng.uniform(0.0, clips[w].duration) for w in which], dtype=np.float32)
    policies = rng.integers(0, 4, size=n).astype(np.uint8)
    max_row = max(clip.num_tracks * clip.num_components for clip in clips)
    out = np.full((n, max_row), -3.0, dtype=np.float32)
    context.decompress_scalar_tracks(np.array(handles, dtype=np.uint32)[which], times, out=out, instance_rounding=policies)
    for i in range(n):
        clip = clips[which[i]]
        expected = ob.oracle_scalar_decompress_tracks(clip.blob, float(times[i]), int(policies[i]))
        used = clip.num_tracks * clip.num_components
        assert helpers.exact(out[i, :used].reshape(clip.num_tracks, clip.num_components), expected)
        assert (out[i, used:] == -3.0).all()        # nothing is written past a clip's own tracks
    for handle in handles:
        context.unregister_clip(handle)


def test_track_types_do_not_cross_entry_points(context):
    scalar_clip = synth.build_scalar_clip(seed=3, track_type=1, num_tracks=6, num_samples=10)
    transform_clip = synth.build_clip(seed=3, num_tracks=6, num_samples=10)
    scalar_handle = context.register_clip(scalar_clip.blob)
    transform_handle = context.register_clip(transform_clip.blob)
    assert context.clip_info(transform_handle).track_type == 12 and context.clip_info(transform_handle).num_components == 12
    before = context.rejected_instance_count()

    # a transform clip through the scalar entry point: rejected, output untouched
    out = np.full((2, 6, 2), 9.0, dtype=np.float32)
    context.decompress_scalar_tracks(np.array([transform_handle, scalar_handle], dtype=np.uint32), np.zeros(2, dtype=np.float32), out=out)
    assert (out[0] == 9.0).all() and helpers.exact(out[1], ob.oracle_scalar_decompress_tracks(scalar_clip.blob, 0.0))

    # a scalar clip through the transform entry points: rejected, output untouched
    poses = np.full((2, 6, 12), 9.0, dtype=np.float32)
    context.decompress_tracks(np.array([scalar_handle, transform_handle], dtype=np.uint32), np.zeros(2, dtype=np.float32), out=poses, num_tracks=6)
    assert (poses[0] == 9.0).all() and helpers.bit_equal(poses[1], ob.oracle_decompress_tracks(transform_clip.blob, 0.0))
    single = np.full((1, 12), 9.0, dtype=np.float32)
    context.decompress_track(np.array([scalar_handle], dtype=np.uint32), np.zeros(1, dtype=np.float32), np.zeros(1, dtype=np.uint32), out=single)
    assert (single == 9.0).all()

    # track index out of range: silently ignored by the reference, counted here
    out = np.full((1, 2), 9.0, dtype=np.float32)
    context.decompress_scalar_track(np.array([scalar_handle], dtype=np.uint32), np.zeros(1, dtype=np.float32), np.array([6], dtype=np.uint32), out=out)
    assert (out == 9.0).all()
    assert context.rejected_instance_count() == before + 4

    # databases are not supported for scalar tracks (decompression.scalar.h:107-108)
    case = helpers.load_database_golden("two_clips_single_chunk")
    database = context.register_database(case["database"], case["bulk_medium"], case["bulk_low"])
    with pytest.raises(runtime.AclHipError):
        context.register_clip_with_database(scalar_clip.blob, database)
    context.unregister_database(database)
    context.unregister_clip(scalar_handle)
    context.unregister_clip(transform_handle)


def test_invalid_and_empty_scalar_clips(context):
    clip = synth.build_scalar_clip(seed=5, track_type=0, num_tracks=10, num_samples=12)
    corrupt = clip.blob.copy()
    corrupt[60] ^= 0x40
    with pytest.raises(runtime.AclHipError):
        context.register_clip(corrupt)                              # hash
    bad_rate = synth.aligned_bytes(clip.blob.size)
    bad_rate[:] = clip.blob
    bad_rate[32 + 20] = 200                                         # first bit rate byte
    with pytest.raises(runtime.AclHipError):
        context.register_clip(bad_rate, check_hash=False)
    truncated = clip.blob[: clip.blob.size - 40].copy()
    with pytest.raises(runtime.AclHipError):
        context.register_clip(truncated, check_hash=False)

    empty = synth.build_scalar_clip(seed=6, track_type=2, num_tracks=0, num_samples=0)
    handle = context.register_clip(empty.blob)
    assert context.clip_info(handle).num_tracks == 0
    out = np.full((3, 4), 5.0, dtype=np.float32)
    context.decompress_scalar_tracks(np.full(3, handle, dtype=np.uint32), np.zeros(3, dtype=np.float32), out=out)
    assert (out == 5.0).all()
    context.unregister_clip(handle)


def test_large_batch_on_device_pointers(context):
    """64k instances of a 256 curve float1f list on device buffers: EVERY instance against the oracle (aclo_scalar_decompress_tracks_batch
    on all host threads) + properties (idempotence, clamping)."""
    import torch
    clip = synth.build_scalar_clip(seed=9, track_type=0, num_tracks=256, num_samples=120, sample_rate=60.0)
    handle = context.register_clip(clip.blob)
    n = 65536
    rng = np.random.default_rng(14)
    times = rng.uniform(-0.2, clip.duration + 0.2, size=n).astype(np.float32)
    d_clips = torch.full((n,), handle, dtype=torch.int32, device="cuda")
    d_times = torch.from_numpy(times).cuda()
    d_values = torch.zeros((n, 256), dtype=torch.float32, device="cuda")
    stream = torch.cuda.current_stream()
    context.decompress_scalar_tracks_batch(d_clips.data_ptr(), d_times.data_ptr(), n, d_values.data_ptr(), 256 * 4, stream=stream.cuda_stream)
    stream.synchronize()
    values = d_values.cpu().numpy()
    assert helpers.exact(values, ob.oracle_scalar_decompress_tracks_batch([clip.blob], np.zeros(n, dtype=np.uint32), times, 256))
    again = torch.zeros_like(d_values)
    context.decompress_scalar_tracks_batch(d_clips.data_ptr(), d_times.data_ptr(), n, again.data_ptr(), 256 * 4, stream=stream.cuda_stream)
    stream.synchronize()
    assert torch.equal(d_values, again)
    clamped = torch.from_numpy(np.clip(times, 0.0, np.float32(clip.duration))).cuda()
    context.decompress_scalar_tracks_batch(d_clips.data_ptr(), clamped.data_ptr(), n, again.data_ptr(), 256 * 4, stream=stream.cuda_stream)
    stream.synchronize()
    assert torch.equal(d_values, again)                            # out of range sample times clamp (decompression.scalar.h:189-190)
    lo, hi = clip.keyframes.min(axis=0)[:, 0], clip.keyframes.max(axis=0)[:, 0]
    assert (values >= lo - 1e-3 * (1 + np.abs(lo))).all() and (values <= hi + 1e-3 * (1 + np.abs(hi))).all()
    context.unregister_clip(handle)


def _oracle_rows(clips, which, times, policies=None, options=None):
    rows = []
    for i in range(len(which)):
        policy = 0 if policies is None else int(policies[i])
        rows.append(ob.oracle_scalar_decompress_tracks(clips[which[i]].blob, float(times[i]), policy, options))
    return rows


@pytest.mark.parametrize("spec", [dict(seed=9, track_type=0, num_tracks=256, num_samples=120, sample_rate=60.0, raw_fraction=0.01),      # the bench list
                                  dict(seed=10, track_type=4, num_tracks=70, num_samples=60, raw_fraction=0.1),
                                  dict(seed=11, track_type=2, num_tracks=300, num_samples=45, wrap=1)])
def test_large_batches_take_groups_of_instances_per_wave(context, spec):
    """from 16 384 instances on a wave decodes 4 consecutive instances (decompress_scalar_tracks_grouped_kernel): groups of one clip
    share the track tables and their round trips, groups of mixed / refused clips fall back -- every value against the oracle"""
    clip = synth.build_scalar_clip(**spec)
    other = synth.build_scalar_clip(seed=spec["seed"] + 50, track_type=spec["track_type"], num_tracks=spec["num_tracks"] - 7, num_samples=33)
    handles = np.array([context.register_clip(clip.blob), context.register_clip(other.blob)], dtype=np.uint32)
    clips = [clip, other]
    rng = np.random.default_rng(spec["seed"])
    n = 16384 + 3                                                        # the last group is partial
    which = np.zeros(n, dtype=np.int64)
    which[5000:5400] = rng.integers(0, 2, size=400)                      # mixed groups in the middle
    which[9000:9016] = 1                                                 # whole groups of the other clip
    durations = np.array([c.duration for c in clips], dtype=np.float32)
    times = (rng.uniform(-0.05, 1.05, size=n).astype(np.float32) * durations[which]).astype(np.float32)
    values = context.decompress_scalar_tracks(handles[which], times)
    check = np.unique(np.concatenate([np.arange(0, n, 97), np.arange(4990, 5410), np.arange(8990, 9030), np.arange(n - 9, n)]))
    expected = _oracle_rows(clips, which[check], times[check])
    for row, i in enumerate(check):
        tracks = clips[which[i]].num_tracks
        assert helpers.exact(values[i, :tracks], expected[row]), f"instance {i}"

    # per instance rounding policies and per track rounding go through the grouped kernel as well
    instance_rounding = rng.integers(0, 4, size=n).astype(np.uint8)
    values = context.decompress_scalar_tracks(handles[which], times, instance_rounding=instance_rounding)
    expected = _oracle_rows(clips, which[check], times[check], policies=instance_rounding[check])
    for row, i in enumerate(check):
        assert helpers.exact(values[i, : clips[which[i]].num_tracks], expected[row]), f"instance {i} (instance rounding)"
    track_rounding = rng.integers(0, 4, size=spec["num_tracks"]).astype(np.uint8)
    params = runtime.default_params(rounding_policy=runtime.ROUND_PER_TRACK, per_track_rounding=1)
    values = context.decompress_scalar_tracks(handles[which], times, params=params, track_rounding=track_rounding)
    options = ob.default_options(per_track_rounding=1)
    options.track_rounding = track_rounding.ctypes.data
    expected = _oracle_rows(clips, which[check], times[check], policies=np.full(n, ob.ROUND_PER_TRACK)[check], options=options)
    for row, i in enumerate(check):
        assert helpers.exact(values[i, : clips[which[i]].num_tracks], expected[row]), f"instance {i} (per track rounding)"

    # an unknown handle inside a group: refused and counted, its neighbours decode (device pointer entry point)
    import torch
    before = context.rejected_instance_count()
    bad = handles[which].copy()
    bad[8] = 123456
    components = clip.num_components
    d_clips = torch.from_numpy(bad.astype(np.int32)).cuda()
    d_times = torch.from_numpy(times).cuda()
    d_values = torch.full((n, spec["num_tracks"] * components), 7.0, dtype=torch.float32, device="cuda")
    context.decompress_scalar_tracks_batch(d_clips.data_ptr(), d_times.data_ptr(), n, d_values.data_ptr(), spec["num_tracks"] * components * 4)
    torch.cuda.synchronize()
    values = d_values.cpu().numpy().reshape(n, spec["num_tracks"], components)
    assert context.rejected_instance_count() == before + 1
    assert np.all(values[8] == 7.0)
    for i in (7, 9, 10, 11):
        assert helpers.exact(values[i, : clip.num_tracks], ob.oracle_scalar_decompress_tracks(clip.blob, float(times[i])))
    for handle in handles:
        context.unregister_clip(int(handle))


def test_float1f_only_registries_take_the_one_float_kernel():
    """while every registered list is float1f the grouped kernel compiled for one float per track runs (half the registers); a wider list
    in the registry switches launches to the kernel for any track type, and back when it leaves -- same values throughout"""
    curves = [synth.build_scalar_clip(seed=60 + k, track_type=0, num_tracks=256 - 31 * k, num_samples=50 + 9 * k, raw_fraction=0.02) for k in range(2)]
    wide = synth.build_scalar_clip(seed=70, track_type=2, num_tracks=40, num_samples=30)
    rng = np.random.default_rng(60)
    n = 16384 + 5
    with runtime.Context(0) as ctx:
        handles = np.array([ctx.register_clip(c.blob) for c in curves], dtype=np.uint32)
        which = (rng.uniform(size=n) < 0.3).astype(np.int64)
        which[4000:8000] = 0
        durations = np.array([c.duration for c in curves], dtype=np.float32)
        times = (rng.uniform(0.0, 1.0, size=n).astype(np.float32) * durations[which]).astype(np.float32)
        check = np.unique(np.concatenate([np.arange(0, n, 61), np.arange(3990, 4010), np.arange(n - 7, n)]))
        expected = _oracle_rows(curves, which[check], times[check])

        def run_and_compare():
            values = ctx.decompress_scalar_tracks(handles[which], times)
            for row, i in enumerate(check):
                assert helpers.exact(values[i, : curves[which[i]].num_tracks], expected[row]), f"instance {i}"

        run_and_compare()                                   # float1f only
        wide_handle = ctx.register_clip(wide.blob)
        run_and_compare()                                   # a float3f list is registered: the kernel for any track type
        mixed_handles = np.concatenate([handles[which[:16380]], np.full(20, wide_handle, dtype=np.uint32)])
        mixed_times = np.concatenate([times[:16380], rng.uniform(0.0, wide.duration, size=20).astype(np.float32)])
        device_values = ctx.decompress_scalar_tracks(mixed_handles, mixed_times)
        for i in range(16380, 16400):
            got = device_values[i].reshape(-1)[: wide.num_tracks * 3].reshape(wide.num_tracks, 3)
            assert helpers.exact(got, ob.oracle_scalar_decompress_tracks(wide.blob, float(mixed_times[i])))
        ctx.unregister_clip(wide_handle)
        run_and_compare()                                   # float1f only again
        assert ctx.rejected_instance_count() == 0
