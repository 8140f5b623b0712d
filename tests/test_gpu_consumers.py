"""Pose consumers fused into the decode (aclhip_decompress_poses_batch) through the C ABI: bit exact against the CPU
oracle's decode -> apply_additive_to_base -> local_to_object_space pipeline, and against the reference's own functions through
the committed fixtures (bit exact for the additive formats; within the tolerance test_pose_consumers_oracle.py explains for object
space, where the reference's x86 arithmetic starts from a hardware estimate). Needs a GPU."""
import numpy as np
import pytest

from acl_amd import runtime, synth
from oracle import bindings as ob
import helpers
from test_pose_consumers_oracle import assert_object_space_close, assert_same_affine_maps

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def context():
    ctx = runtime.Context(0)
    yield ctx
    ctx.close()


def random_hierarchy(rng, num_tracks, parent_span, extra_roots=0):
    parents = np.zeros(num_tracks, dtype=np.uint32)
    parents[0] = runtime.NO_PARENT
    for i in range(1, num_tracks):
        parents[i] = rng.integers(max(0, i - parent_span), i)
    if extra_roots and num_tracks > 1:
        parents[rng.choice(np.arange(1, num_tracks), size=min(extra_roots, num_tracks - 1), replace=False)] = runtime.NO_PARENT
    return parents


def oracle_poses(blobs, clip_indices, times, rounding=0, options=None):
    return np.stack([ob.oracle_decompress_tracks(blobs[c], float(t), rounding, options) for c, t in zip(clip_indices, times)])


@pytest.mark.parametrize("name", helpers.consumer_golden_cases())
def test_matches_reference_fixtures(context, name):
    case = helpers.load_consumer_golden(name)
    additive, base = context.register_clip(case["additive_blob"]), context.register_clip(case["base_blob"])
    context.set_clip_hierarchy(additive, case["parents"])
    n = case["times"].shape[0]
    additive_handles, base_handles = np.full(n, additive, dtype=np.uint32), np.full(n, base, dtype=np.uint32)
    additive_times, base_times = case["times"][:, 0].copy(), case["times"][:, 1].copy()
    base_poses = context.decompress_tracks(base_handles, base_times)
    for additive_format in range(4):
        kwargs = dict(additive_format=additive_format)
        if additive_format != runtime.ADDITIVE_NONE:
            kwargs.update(base_clips=base_handles, base_sample_times=base_times)
        local = context.decompress_poses(additive_handles, additive_times, **kwargs)
        assert helpers.bit_equal(local, case["local"][additive_format]), (name, additive_format)
        object_space = context.decompress_poses(additive_handles, additive_times, object_space=True, **kwargs)
        assert_object_space_close(object_space, case["object_space"][additive_format])
        # the base as a pose buffer in HBM instead of a clip instance decoded by the same wave: same bits
        if additive_format != runtime.ADDITIVE_NONE:
            from_buffer = context.decompress_poses(additive_handles, additive_times, additive_format=additive_format, base_poses=base_poses, object_space=True)
            assert helpers.exact(from_buffer, object_space)
        # and bit exact against the oracle pipeline
        for index in range(n):
            additive_pose = ob.oracle_decompress_tracks(case["additive_blob"], float(additive_times[index]))
            base_pose = ob.oracle_decompress_tracks(case["base_blob"], float(base_times[index]))
            expected_local = ob.oracle_apply_additive_to_base(additive_format, base_pose, additive_pose)
            assert helpers.exact(local[index], expected_local)
            assert helpers.exact(object_space[index], ob.oracle_local_to_object_space(case["parents"], expected_local))
    context.unregister_clip(additive)
    context.unregister_clip(base)
    assert context.rejected_instance_count() == 0


CLIP_SHAPES = {
    "one_bone": dict(seed=301, num_tracks=1, num_samples=9),
    "small_scale": dict(seed=302, num_tracks=17, num_samples=33, has_scale=1, scale_default=0.3),
    "window_boundary_106": dict(seed=303, num_tracks=106, num_samples=40, has_scale=1),
    "window_boundary_107": dict(seed=304, num_tracks=107, num_samples=40, has_scale=1),
    "crowd_rig_1200": dict(seed=305, num_tracks=1200, num_samples=12, has_scale=1, scale_default=0.5),
    "stripped_wrap": dict(seed=306, num_tracks=50, num_samples=100, strip_keyframes=1, wrap=1, has_scale=1),
    "mostly_default": dict(seed=307, num_tracks=64, num_samples=20, rotation_default=0.6, translation_default=0.6, has_scale=1, scale_default=0.8),
}


@pytest.mark.parametrize("name", sorted(CLIP_SHAPES))
def test_matches_oracle_on_synthetic_clips(name):
    """Every rounding policy and looping policy, mixed clips in one batch, per instance rounding; a context of its own because
    the LDS image size follows the largest registered clip."""
    rng = np.random.default_rng(CLIP_SHAPES[name]["seed"])
    spec = CLIP_SHAPES[name]
    clips = [synth.build_clip(**spec), synth.build_clip(**dict(spec, seed=spec["seed"] + 1000, num_samples=spec["num_samples"] + 7))]
    blobs = [c.blob for c in clips]
    num_tracks = spec["num_tracks"]
    parents = random_hierarchy(rng, num_tracks, parent_span=max(1, num_tracks // 8), extra_roots=2 if num_tracks > 8 else 0)
    with runtime.Context(0) as context:
        handles = [context.register_clip(b) for b in blobs]
        for handle in handles:
            context.set_clip_hierarchy(handle, parents)
        n = 24 if num_tracks < 500 else 6
        which = rng.integers(0, 2, size=n)
        base_which = rng.integers(0, 2, size=n)
        times = np.array([rng.uniform(-0.05, clips[c].duration + 0.05) for c in which], dtype=np.float32)
        base_times = np.array([rng.uniform(-0.05, clips[c].duration + 0.05) for c in base_which], dtype=np.float32)
        clip_handles = np.array([handles[c] for c in which], dtype=np.uint32)
        base_handles = np.array([handles[c] for c in base_which], dtype=np.uint32)
        for rounding, looping, additive_format in ((0, 2, 1), (1, 0, 2), (2, 1, 3), (3, 2, 0), (0, 1, 1)):
            params = runtime.default_params(rounding_policy=rounding, looping_policy=looping)
            options = ob.default_options(looping_policy=looping)
            kwargs = dict(additive_format=additive_format, params=params, object_space=True)
            if additive_format != 0:
                kwargs.update(base_clips=base_handles, base_sample_times=base_times)
            got = context.decompress_poses(clip_handles, times, **kwargs)
            additive_poses = oracle_poses(blobs, which, times, rounding, options)
            base_poses = oracle_poses(blobs, base_which, base_times, rounding, options)
            for i in range(n):
                local = ob.oracle_apply_additive_to_base(additive_format, base_poses[i], additive_poses[i])
                assert helpers.exact(got[i], ob.oracle_local_to_object_space(parents, local)), (name, rounding, looping, additive_format, i)
        # per instance rounding policies reach both the instance and its base
        instance_rounding = rng.integers(0, 4, size=n).astype(np.uint8)
        got = context.decompress_poses(clip_handles, times, additive_format=2, base_clips=base_handles, base_sample_times=base_times, instance_rounding=instance_rounding)
        for i in range(n):
            a = ob.oracle_decompress_tracks(blobs[which[i]], float(times[i]), int(instance_rounding[i]))
            b = ob.oracle_decompress_tracks(blobs[base_which[i]], float(base_times[i]), int(instance_rounding[i]))
            assert helpers.exact(got[i], ob.oracle_apply_additive_to_base(2, b, a))
        assert context.rejected_instance_count() == 0


def test_no_consumers_equals_decompress_tracks(context):
    clip = synth.build_clip(seed=77, num_tracks=90, num_samples=50, has_scale=1)
    handle = context.register_clip(clip.blob)
    times = np.linspace(0.0, clip.duration, 40, dtype=np.float32)
    handles = np.full(times.size, handle, dtype=np.uint32)
    assert helpers.exact(context.decompress_poses(handles, times), context.decompress_tracks(handles, times))
    context.unregister_clip(handle)


def test_transforms_that_meet_a_negative_scale_are_counted(context):
    """rtm::qvv_mul composes matrices when a scale component is negative (mirrored rigs); the kernels do the same and count such
    transforms (aclhip_get_negative_scale_count) -- `relative` additive onto mirrored base poses, then object space"""
    clip = synth.build_clip(seed=78, num_tracks=24, num_samples=30, has_scale=1)
    handle = context.register_clip(clip.blob)
    rng = np.random.default_rng(5)
    parents = random_hierarchy(rng, 24, 3)
    context.set_clip_hierarchy(handle, parents)
    n = 9
    times = np.linspace(0.0, clip.duration, n, dtype=np.float32)
    handles = np.full(n, handle, dtype=np.uint32)
    base = np.zeros((n, 24, 12), dtype=np.float32)
    base[..., 3] = 1.0
    base[..., 8:11] = 1.0
    before = context.negative_scale_count()
    context.decompress_poses(handles, times, additive_format=runtime.ADDITIVE_RELATIVE, base_poses=base)
    local = context.decompress_poses(handles, times)
    assert np.all(local[..., 8:11] > 0.0)                                      # the clip itself holds positive scales
    assert context.negative_scale_count() == before                             # nothing mirrored: nothing counted
    mirrored = base.copy()
    mirrored[2, 5, 8] = -1.0                                                    # one mirrored base transform of one instance ...
    mirrored[7, :, 9] = -1.0                                                    # ... and a whole mirrored base pose
    context.decompress_poses(handles, times, additive_format=runtime.ADDITIVE_RELATIVE, base_poses=mirrored)
    assert context.negative_scale_count() == before + 1 + 24
    # object space: every transform below a mirrored one multiplies with a negative scale; roots never do
    before = context.negative_scale_count()
    context.decompress_poses(handles[:1], times[7:8], additive_format=runtime.ADDITIVE_RELATIVE, base_poses=mirrored[7:8], object_space=True)
    num_roots = int(np.sum((parents == runtime.NO_PARENT) | (np.arange(24) == 0)))
    assert context.negative_scale_count() == before + 24 + (24 - num_roots)
    context.unregister_clip(handle)


def test_mirrored_rig_in_object_space_is_the_fp64_matrix_chain(context):
    """A mirrored 100-bone rig: negative scales take rtm::qvv_mul's route through 3x4 matrices (aclhip_device.h:
    qvv_mul_through_matrices). Bit for bit the oracle, and -- independently of any reading of RTM -- the same affine map per bone as
    the fp64 product of the local matrices down its chain of parents (scales of one magnitude per bone keep the chain free of shear)."""
    clip = synth.build_clip(seed=91, num_tracks=100, num_samples=61)                      # no scale sub-tracks: every local scale is 1
    handle = context.register_clip(clip.blob)
    parents = synth.humanoid_hierarchy(100)
    context.set_clip_hierarchy(handle, parents)
    rng = np.random.default_rng(91)
    n = 64
    times = rng.uniform(0.0, clip.duration, size=n).astype(np.float32)
    handles = np.full(n, handle, dtype=np.uint32)
    # additive0 onto a base of identity rotations, zero translations and scales of +-m: the local pose is the clip's with mirrored scales
    base = np.zeros((n, 100, 12), dtype=np.float32)
    base[..., 3] = 1.0
    magnitude = rng.uniform(0.8, 1.25, size=(n, 100, 1))
    base[..., 8:11] = (magnitude * np.where(rng.uniform(size=(n, 100, 3)) < 0.25, -1.0, 1.0)).astype(np.float32)
    before = context.negative_scale_count()
    got = context.decompress_poses(handles, times, additive_format=runtime.ADDITIVE_ADDITIVE0, base_poses=base, object_space=True)
    assert context.negative_scale_count() > before
    for i in range(n):
        local = ob.oracle_apply_additive_to_base(runtime.ADDITIVE_ADDITIVE0, base[i], ob.oracle_decompress_tracks(clip.blob, float(times[i])))
        assert (local[:, 8:11] < 0.0).any()
        assert helpers.bit_equal(got[i], ob.oracle_local_to_object_space(parents, local)), i
        assert_same_affine_maps(got[i], parents, local)
    # the `relative` format is a qvv_mul too: mirrored additive onto mirrored base, per transform, against the fp64 matrix product
    relative = context.decompress_poses(handles, times, additive_format=runtime.ADDITIVE_RELATIVE, base_poses=base)
    for i in range(0, n, 7):
        additive = ob.oracle_decompress_tracks(clip.blob, float(times[i]))
        assert helpers.bit_equal(relative[i], ob.oracle_apply_additive_to_base(runtime.ADDITIVE_RELATIVE, base[i], additive)), i
    context.unregister_clip(handle)


def test_registered_clips_with_negative_scales():
    """Clips whose OWN scale sub-tracks are negative (constant and animated): registration notices (the pose consumers then run with
    the matrix route compiled in), every consumer path stays bit exact against the oracle, transforms are counted; a context without
    such clips counts nothing."""
    context = runtime.Context(0)
    try:
        mirrored = synth.build_clip(seed=92, num_tracks=100, num_samples=45, has_scale=1, scale_default=0.3, scale_constant=0.3, mirrored_scale_fraction=0.3)
        plain = synth.build_clip(seed=93, num_tracks=100, num_samples=45, has_scale=1)
        parents = synth.humanoid_hierarchy(100)
        rng = np.random.default_rng(92)
        n = 48
        times = rng.uniform(0.0, mirrored.duration, size=n).astype(np.float32)

        plain_handle = context.register_clip(plain.blob)
        context.set_clip_hierarchy(plain_handle, parents)
        context.decompress_poses(np.full(n, plain_handle, dtype=np.uint32), times, object_space=True)
        assert context.negative_scale_count() == 0

        handle = context.register_clip(mirrored.blob)
        context.set_clip_hierarchy(handle, parents)
        handles = np.full(n, handle, dtype=np.uint32)
        local = oracle_poses([mirrored.blob], np.zeros(n, dtype=np.int64), times)
        assert (local[..., 8:11] < 0.0).any()
        got = context.decompress_poses(handles, times, object_space=True)
        assert context.negative_scale_count() > 0
        for i in range(n):
            assert helpers.bit_equal(got[i], ob.oracle_local_to_object_space(parents, local[i])), i
        # a mirrored clip as the BASE of a relative additive clip (second wave, second image), then object space
        base_times = rng.uniform(0.0, mirrored.duration, size=n).astype(np.float32)
        got = context.decompress_poses(np.full(n, plain_handle, dtype=np.uint32), times, additive_format=runtime.ADDITIVE_RELATIVE,
                                       base_clips=handles, base_sample_times=base_times, object_space=True)
        additive = oracle_poses([plain.blob], np.zeros(n, dtype=np.int64), times)
        base = oracle_poses([mirrored.blob], np.zeros(n, dtype=np.int64), base_times)
        for i in range(n):
            combined = ob.oracle_apply_additive_to_base(runtime.ADDITIVE_RELATIVE, base[i], additive[i])
            assert helpers.bit_equal(got[i], ob.oracle_local_to_object_space(parents, combined)), i
        # additive1 onto the mirrored base by the instance's own wave (the fused path), then object space
        got = context.decompress_poses(np.full(n, plain_handle, dtype=np.uint32), times, additive_format=runtime.ADDITIVE_ADDITIVE1,
                                       base_clips=handles, base_sample_times=base_times, object_space=True)
        for i in range(n):
            combined = ob.oracle_apply_additive_to_base(runtime.ADDITIVE_ADDITIVE1, base[i], additive[i])
            assert helpers.bit_equal(got[i], ob.oracle_local_to_object_space(parents, combined)), i
        assert context.rejected_instance_count() == 0
    finally:
        context.close()


def test_refused_instances_and_arguments(context):
    rng = np.random.default_rng(3)
    with_hierarchy = synth.build_clip(seed=81, num_tracks=20, num_samples=20)
    without = synth.build_clip(seed=82, num_tracks=20, num_samples=20)
    other_size = synth.build_clip(seed=83, num_tracks=21, num_samples=20)
    scalar = synth.build_scalar_clip(seed=84, track_type=0, num_tracks=20, num_samples=10)
    h_with, h_without, h_other, h_scalar = (context.register_clip(c.blob) for c in (with_hierarchy, without, other_size, scalar))
    parents = random_hierarchy(rng, 20, 4)
    context.set_clip_hierarchy(h_with, parents)
    before = context.rejected_instance_count()

    # object space without a hierarchy, unknown handle, scalar clip: refused, counted, output untouched
    handles = np.array([h_with, h_without, 12345, h_scalar], dtype=np.uint32)
    times = np.zeros(4, dtype=np.float32)
    out = np.full((4, 20, 12), 7.0, dtype=np.float32)
    context.decompress_poses(handles, times, object_space=True, out=out, num_tracks=20)
    assert context.rejected_instance_count() == before + 3
    assert np.all(out[1:] == 7.0) and not np.any(out[0] == 7.0)
    # without object space the clip without hierarchy is fine
    out = context.decompress_poses(np.array([h_without], dtype=np.uint32), times[:1])
    assert helpers.exact(out[0], ob.oracle_decompress_tracks(without.blob, 0.0))

    # a base clip with another number of tracks, an unknown base clip
    before = context.rejected_instance_count()
    out = np.full((2, 21, 12), 7.0, dtype=np.float32)
    context.decompress_poses(np.array([h_with, h_with], dtype=np.uint32), times[:2], additive_format=1, base_clips=np.array([h_other, 999], dtype=np.uint32),
                             base_sample_times=times[:2], out=out, num_tracks=21)
    assert context.rejected_instance_count() == before + 2 and np.all(out == 7.0)

    # hierarchies must be sorted parent first, sized like the clip, and belong to a transform clip
    for bad in (np.array([0] * 19, dtype=np.uint32), np.arange(1, 21, dtype=np.uint32)):
        with pytest.raises(runtime.AclHipError) as error:
            context.set_clip_hierarchy(h_with, bad)
        assert error.value.status == runtime.ERROR_INVALID_ARGUMENT
    with pytest.raises(runtime.AclHipError):
        context.set_clip_hierarchy(h_scalar, parents)
    with pytest.raises(runtime.AclHipError) as error:
        context.set_clip_hierarchy(4242, parents)
    assert error.value.status == runtime.ERROR_UNKNOWN_CLIP

    # settings a consumer cannot work with
    handles = np.array([h_with], dtype=np.uint32)
    for overrides in (dict(per_track_rounding=1), dict(normalization=runtime.NORMALIZE_ALWAYS), dict(default_rotation_mode=runtime.DEFAULT_SKIPPED)):
        with pytest.raises(runtime.AclHipError) as error:
            context.decompress_poses(handles, times[:1], params=runtime.default_params(**overrides))
        assert error.value.status == runtime.ERROR_INVALID_ARGUMENT
    with pytest.raises(runtime.AclHipError):
        context.decompress_poses(handles, times[:1], additive_format=7)
    with pytest.raises(runtime.AclHipError):       # an additive format without any base
        context.decompress_poses(handles, times[:1], additive_format=1)

    # replacing a hierarchy takes effect
    chain = np.concatenate([[runtime.NO_PARENT], np.arange(0, 19)]).astype(np.uint32)
    context.set_clip_hierarchy(h_with, chain)
    out = context.decompress_poses(handles, times[:1], object_space=True)
    assert helpers.exact(out[0], ob.oracle_local_to_object_space(chain, ob.oracle_decompress_tracks(with_hierarchy.blob, 0.0)))
    for handle in (h_with, h_without, h_other, h_scalar):
        context.unregister_clip(handle)


def test_pose_too_large_for_lds():
    with runtime.Context(0) as context:
        big = synth.build_clip(seed=91, num_tracks=1800, num_samples=4)
        handle = context.register_clip(big.blob)
        handles, times = np.array([handle], dtype=np.uint32), np.zeros(1, dtype=np.float32)
        # one image fits (1800 * 48 = 86 400 bytes), two do not
        out = context.decompress_poses(handles, times)
        assert helpers.exact(out[0], ob.oracle_decompress_tracks(big.blob, 0.0))
        with pytest.raises(runtime.AclHipError) as error:
            context.decompress_poses(handles, times, additive_format=1, base_clips=handles, base_sample_times=times)
        assert error.value.status == runtime.ERROR_INVALID_ARGUMENT


def test_full_size_batch_properties(context):
    """65 536 instances (BASELINE's batch): EVERY instance of the fused kernel against the oracle's decode -> additive -> object space
    pipeline (aclo_decompress_poses_batch on all host threads), for object space alone and for additive1 onto a base clip instance +
    object space -- the two pose consumer workloads bench.py times."""
    import torch
    clip = synth.build_clip(seed=7, num_tracks=100, num_samples=301)
    handle = context.register_clip(clip.blob)
    rng = np.random.default_rng(0)
    parents = random_hierarchy(rng, 100, 10)
    context.set_clip_hierarchy(handle, parents)
    n = 65536
    times = torch.from_numpy(rng.uniform(0.0, clip.duration, size=n).astype(np.float32)).cuda()
    handles = torch.full((n,), handle, dtype=torch.int32).cuda()
    poses = torch.zeros((n, 100, 12), dtype=torch.float32, device="cuda")
    consumers = runtime.PoseConsumers()
    consumers.object_space = 1
    context.decompress_poses_batch(handles.data_ptr(), times.data_ptr(), n, poses.data_ptr(), 4800, consumers)
    torch.cuda.synchronize()
    got = poses.cpu().numpy()
    host_times = times.cpu().numpy()
    expected = ob.oracle_decompress_poses_batch([clip.blob], np.zeros(n, dtype=np.uint32), host_times, 100, parent_indices=parents)
    assert helpers.bit_equal(got, expected)
    # object space rotations stay normalized, whatever the depth
    lengths = np.linalg.norm(got[:, :, 0:4], axis=2)
    assert np.abs(lengths - 1.0).max() < 1.0e-5

    # additive1 onto a base clip instance decoded by the instance's own wave, then object space: every instance again
    additive = synth.build_clip(seed=12, num_tracks=100, num_samples=121, rotation_constant=0.5, translation_constant=0.8)
    additive_handle = context.register_clip(additive.blob)
    context.set_clip_hierarchy(additive_handle, parents)
    additive_times = rng.uniform(0.0, additive.duration, size=n).astype(np.float32)
    consumers.additive_format = runtime.ADDITIVE_ADDITIVE1
    consumers.base_clips = handles.data_ptr()
    consumers.base_sample_times = times.data_ptr()
    d_additive_clips = torch.full((n,), additive_handle, dtype=torch.int32).cuda()
    d_additive_times = torch.from_numpy(additive_times).cuda()
    context.decompress_poses_batch(d_additive_clips.data_ptr(), d_additive_times.data_ptr(), n, poses.data_ptr(), 4800, consumers)
    torch.cuda.synchronize()
    expected = ob.oracle_decompress_poses_batch([clip.blob, additive.blob], np.ones(n, dtype=np.uint32), additive_times, 100, additive_format=runtime.ADDITIVE_ADDITIVE1,
                                                base_clip_indices=np.zeros(n, dtype=np.uint32), base_sample_times=host_times, parent_indices=parents)
    assert helpers.bit_equal(poses.cpu().numpy(), expected)
    context.unregister_clip(additive_handle)
    context.unregister_clip(handle)
    assert context.rejected_instance_count() >= 0


def test_database_bound_clips_follow_the_tiers_in_object_space():
    """pose consumers decode through the same seek as the pose kernels: a clip bound to a compressed_database picks its keyframes
    from whatever tiers are resident -- object space of the reference's golden poses after every stream_in / stream_out request"""
    case = helpers.load_database_golden("three_clips_4k_chunks")
    rng = np.random.default_rng(17)
    with runtime.Context(0) as context:
        database = context.register_database(case["database"], case["bulk_medium"], case["bulk_low"])
        clips = [context.register_clip_with_database(clip, database) for clip in case["clips"]]
        num_tracks = [ob.oracle().aclo_num_tracks(clip.ctypes.data) for clip in case["clips"]]
        parents = [random_hierarchy(rng, n, parent_span=6, extra_roots=1) for n in num_tracks]
        for clip, clip_parents in zip(clips, parents):
            context.set_clip_hierarchy(clip, clip_parents)
        num_times = case["times"].shape[1]
        max_tracks = case["poses"].shape[4]
        handles = np.repeat(np.array(clips, dtype=np.uint32), num_times)
        times = case["times"].reshape(-1)

        def check(state):
            poses = context.decompress_poses(handles, times, object_space=True, num_tracks=max_tracks)
            for c in range(len(clips)):
                for i in range(num_times):
                    local = case["poses"][state, c, 0, i, :num_tracks[c]].copy()      # policy 0 = none
                    local[:, 7] = 0.0
                    local[:, 11] = 0.0
                    expected = ob.oracle_local_to_object_space(parents[c], local)
                    assert helpers.exact(poses[c * num_times + i, :num_tracks[c]], expected), (state, c, i)

        assert int(case["policies"][0]) == 0
        check(0)
        for state, (tier, num_chunks, stream_in) in enumerate(case["ops"]):
            (context.database_stream_in if stream_in else context.database_stream_out)(database, int(tier), int(num_chunks))
            check(state + 1)
        assert context.rejected_instance_count() == 0


def test_mixed_skeletons_inside_one_workgroup():
    """instances of one workgroup usually share a skeleton and walk from one LDS copy of its schedule; when they do not, every
    instance follows its own (different hierarchies, different sizes, clips without one next to clips with one)"""
    rng = np.random.default_rng(23)
    with runtime.Context(0) as context:
        specs = [dict(seed=601, num_tracks=50, num_samples=30), dict(seed=602, num_tracks=50, num_samples=30, has_scale=1),
                 dict(seed=603, num_tracks=23, num_samples=30), dict(seed=604, num_tracks=50, num_samples=20)]
        clips = [synth.build_clip(**spec) for spec in specs]
        handles = [context.register_clip(c.blob) for c in clips]
        parents = [random_hierarchy(rng, 50, 5), random_hierarchy(rng, 50, 20, extra_roots=2), random_hierarchy(rng, 23, 3), None]
        parents[3] = parents[0].copy()              # same skeleton as clip 0: one shared schedule image
        for handle, clip_parents in zip(handles, parents):
            context.set_clip_hierarchy(handle, clip_parents)
        n = 257
        which = rng.integers(0, len(clips), size=n)
        times = np.array([rng.uniform(0.0, clips[c].duration) for c in which], dtype=np.float32)
        got = context.decompress_poses(np.array([handles[c] for c in which], dtype=np.uint32), times, object_space=True, num_tracks=50)
        for i in range(n):
            c = which[i]
            local = ob.oracle_decompress_tracks(clips[c].blob, float(times[i]))
            assert helpers.exact(got[i, :clips[c].num_tracks], ob.oracle_local_to_object_space(parents[c], local)), (i, c)
        # dropping one user of the shared image leaves the other intact
        context.unregister_clip(handles[0])
        got = context.decompress_poses(np.full(8, handles[3], dtype=np.uint32), times[:8] * 0.0, object_space=True, num_tracks=50)
        assert helpers.exact(got[0], ob.oracle_local_to_object_space(parents[3], ob.oracle_decompress_tracks(clips[3].blob, 0.0)))
        assert context.rejected_instance_count() == 0


@pytest.mark.parametrize("num_tracks", [1, 7, 100, 104, 105, 333])
def test_poses_without_scale_keep_no_scale_in_lds(num_tracks):
    """no registered clip has a scale other than 1: the object space images hold rotation | translation only (half as many poses again
    per CU). Same bits as the oracle and as the three-quad images (ACLHIP_CONSUMER_KEEP_SCALE); a clip WITH scale in the registry
    switches the launch back, and mixes of both kinds decode together."""
    import os
    rng = np.random.default_rng(num_tracks)
    clips = [synth.build_clip(seed=800 + num_tracks + k, num_tracks=num_tracks, num_samples=30 + 11 * k) for k in range(2)]
    parents = random_hierarchy(rng, num_tracks, parent_span=max(1, num_tracks // 6), extra_roots=1 if num_tracks > 4 else 0)
    with runtime.Context(0) as context:
        handles = [context.register_clip(c.blob) for c in clips]
        for handle in handles:
            context.set_clip_hierarchy(handle, parents)
        n = 37
        which = rng.integers(0, 2, size=n)
        times = np.array([rng.uniform(0.0, clips[w].duration) for w in which], dtype=np.float32)
        clip_handles = np.array([handles[w] for w in which], dtype=np.uint32)
        expected = np.stack([ob.oracle_local_to_object_space(parents, ob.oracle_decompress_tracks(clips[w].blob, float(t))) for w, t in zip(which, times)])

        two_quads = context.decompress_poses(clip_handles, times, object_space=True)
        assert helpers.exact(two_quads, expected)
        assert np.array_equal(two_quads[:, :, 8:12], np.broadcast_to(np.array([1.0, 1.0, 1.0, 0.0], dtype=np.float32), (n, num_tracks, 4)))
        os.environ["ACLHIP_CONSUMER_KEEP_SCALE"] = "1"
        try:
            three_quads = context.decompress_poses(clip_handles, times, object_space=True)
        finally:
            del os.environ["ACLHIP_CONSUMER_KEEP_SCALE"]
        assert helpers.exact(three_quads, two_quads)

        # a clip with scale joins the registry: launches keep the scales again, for every clip
        scaled = synth.build_clip(seed=900 + num_tracks, num_tracks=num_tracks, num_samples=20, has_scale=1)
        scaled_handle = context.register_clip(scaled.blob)
        context.set_clip_hierarchy(scaled_handle, parents)
        mixed_handles = np.concatenate([clip_handles, np.full(5, scaled_handle, dtype=np.uint32)])
        mixed_times = np.concatenate([times, rng.uniform(0.0, scaled.duration, size=5).astype(np.float32)])
        mixed = context.decompress_poses(mixed_handles, mixed_times, object_space=True)
        assert helpers.exact(mixed[:n], expected)
        for i in range(5):
            local = ob.oracle_decompress_tracks(scaled.blob, float(mixed_times[n + i]))
            assert helpers.exact(mixed[n + i], ob.oracle_local_to_object_space(parents, local))
        # ... and leaves it again
        context.unregister_clip(scaled_handle)
        assert helpers.exact(context.decompress_poses(clip_handles, times, object_space=True), expected)
        assert context.rejected_instance_count() == 0


# ---- ACLHIP_CONSUMERS_FAST: the opt-in arithmetic (1 ulp square roots, fused multiply-adds) against the bit exact default -----------------
FAST_ROTATION_TOLERANCE = 2.0e-6        # per component of a unit rotation
FAST_TRANSLATION_TOLERANCE = 2.0e-6     # relative to max(1, largest |translation| of the pose)


def assert_fast_close(fast, exact):
    rotation = float(np.abs(fast[..., 0:4] - exact[..., 0:4]).max())
    extent = max(1.0, float(np.abs(exact[..., 4:7]).max()))
    translation = float(np.abs(fast[..., 4:7] - exact[..., 4:7]).max()) / extent
    scale = float(np.abs(fast[..., 8:11] - exact[..., 8:11]).max()) / max(1.0, float(np.abs(exact[..., 8:11]).max()))
    assert rotation <= FAST_ROTATION_TOLERANCE and translation <= FAST_TRANSLATION_TOLERANCE and scale <= FAST_TRANSLATION_TOLERANCE, (rotation, translation, scale)
    assert np.all(fast[..., 7] == 0.0) and np.all(fast[..., 11] == 0.0)
    return rotation, translation


@pytest.mark.parametrize("name", ["small_scale", "window_boundary_107", "crowd_rig_1200", "stripped_wrap", "mostly_default"])
def test_fast_arithmetic_stays_within_its_tolerance_of_the_bit_exact_kernels(name):
    rng = np.random.default_rng(CLIP_SHAPES[name]["seed"] + 5)
    spec = CLIP_SHAPES[name]
    clips = [synth.build_clip(**spec), synth.build_clip(**dict(spec, seed=spec["seed"] + 1000, num_samples=spec["num_samples"] + 7))]
    num_tracks = spec["num_tracks"]
    parents = random_hierarchy(rng, num_tracks, parent_span=max(1, num_tracks // 8), extra_roots=2 if num_tracks > 8 else 0)
    with runtime.Context(0) as context:
        handles = [context.register_clip(c.blob) for c in clips]
        for handle in handles:
            context.set_clip_hierarchy(handle, parents)
        n = 32 if num_tracks < 500 else 8
        which, base_which = rng.integers(0, 2, size=n), rng.integers(0, 2, size=n)
        times = np.array([rng.uniform(0.0, clips[c].duration) for c in which], dtype=np.float32)
        base_times = np.array([rng.uniform(0.0, clips[c].duration) for c in base_which], dtype=np.float32)
        clip_handles, base_handles = np.array([handles[c] for c in which], dtype=np.uint32), np.array([handles[c] for c in base_which], dtype=np.uint32)
        base_poses = context.decompress_tracks(base_handles, base_times)
        for additive_format, base_as_buffer in ((0, False), (1, False), (2, False), (3, False), (3, True)):
            kwargs = dict(additive_format=additive_format, object_space=True)
            if additive_format != 0 and base_as_buffer:
                kwargs.update(base_poses=base_poses)
            elif additive_format != 0:
                kwargs.update(base_clips=base_handles, base_sample_times=base_times)
            exact = context.decompress_poses(clip_handles, times, **kwargs)
            fast = context.decompress_poses(clip_handles, times, flags=runtime.CONSUMERS_FAST, **kwargs)
            assert_fast_close(fast, exact)
            assert not helpers.exact(fast, exact) or num_tracks == 1            # (it IS another arithmetic)
        # local space output ignores the flag: the default, bit exact kernels run
        assert helpers.exact(context.decompress_poses(clip_handles, times, flags=runtime.CONSUMERS_FAST), context.decompress_poses(clip_handles, times))
        assert context.rejected_instance_count() == 0


def test_fast_object_space_of_the_humanoid_is_the_fp64_chain(context):
    """the 100-bone character of the bench workloads (13 levels deep): fast object space against the fp64 product of the local matrices
    down every bone's chain of parents, and against the bit exact kernel"""
    clip = synth.build_clip(seed=2, num_tracks=100, num_samples=301, sample_rate=30.0)
    handle = context.register_clip(clip.blob)
    parents = synth.humanoid_hierarchy(100)
    context.set_clip_hierarchy(handle, parents)
    rng = np.random.default_rng(17)
    n = 256
    times = rng.uniform(0.0, clip.duration, size=n).astype(np.float32)
    handles = np.full(n, handle, dtype=np.uint32)
    exact = context.decompress_poses(handles, times, object_space=True)
    fast = context.decompress_poses(handles, times, object_space=True, flags=runtime.CONSUMERS_FAST)
    rotation, translation = assert_fast_close(fast, exact)
    print(f"fast vs bit exact, 100-bone humanoid: rotations {rotation:.3e}, translations {translation:.3e} of the extent")
    for i in range(0, n, 16):
        assert_same_affine_maps(fast[i], parents, ob.oracle_decompress_tracks(clip.blob, float(times[i])), tolerance=2.0e-6)
    assert np.abs(np.linalg.norm(fast[..., 0:4].astype(np.float64), axis=-1) - 1.0).max() <= 1.0e-6
    context.unregister_clip(handle)
