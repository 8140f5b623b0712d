"""aclhip_output_desc: the output side of the track_writer protocol (core/track_writer.h:161-216) -- compact pose layouts and
skipped sub-track kinds (skip_all_rotations / translations / scales, :181-183). Same values as the QVV48 decode, compared through
the layout against the oracle; what a writer skips keeps the bytes the caller had there. Needs a GPU."""
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
SHAPES = ["cmu_100", "scale_37", "stripped_wrap_scale", "raw_and_constant_rates", "cinematic_300", "three_full_windows_320", "crowd_rig_1200", "all_default",
          "two_samples_three_tracks", "one_sample"]


def launch(context, handles, times, layout, skip=(0, 0, 0), rows=None, num_rows=None, max_tracks=None, params=None, skip_tracks=None):
    layout_id, bytes_per_track = runtime.LAYOUTS[layout]
    stride = (max_tracks * bytes_per_track + 15) // 16 * 16
    num_rows = num_rows or handles.size
    d_handles = torch.from_numpy(handles.astype(np.int32)).cuda()
    d_times = torch.from_numpy(times).cuda()
    poses = torch.full((num_rows, stride // 4), FILL, dtype=torch.float32, device="cuda")
    output = runtime.OutputDesc()
    output.layout = layout_id
    output.skip_rotations, output.skip_translations, output.skip_scales = skip
    if rows is not None:
        d_rows = torch.from_numpy(rows.astype(np.int32)).cuda()
        output.rows = d_rows.data_ptr()
    if skip_tracks is not None:
        d_skip_tracks = torch.from_numpy(skip_tracks.astype(np.uint8)).cuda()
        output.skip_tracks = d_skip_tracks.data_ptr()
    context.decompress_tracks_batch_out(d_handles.data_ptr(), d_times.data_ptr(), handles.size, poses.data_ptr(), stride, output, params=params)
    torch.cuda.synchronize()
    return poses.cpu().numpy()[:, : max_tracks * bytes_per_track // 4].reshape(num_rows, max_tracks, bytes_per_track // 4)


def expected_through_layout(oracle_poses, layout, skip):
    """[n, tracks, 12] oracle poses -> what the buffer must hold: skipped kinds (and the scale lanes QV32 does not have) untouched"""
    layout_id = runtime.LAYOUTS[layout][0]
    width = runtime.LAYOUTS[layout][1] // 4
    background = np.full(oracle_poses.shape[:-1] + (width,), FILL, dtype=np.float32)
    return runtime.relayout_pose(oracle_poses, layout_id, skip=tuple(bool(s) for s in skip), into=background)


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


@pytest.mark.parametrize("name", SHAPES)
@pytest.mark.parametrize("layout", ["qvv48", "qvv40", "qv32"])
def test_layouts_hold_the_oracle_values(context, name, layout):
    clip = synth.build_clip(**CLIP_SPECS[name])
    handle = context.register_clip(clip.blob)
    rng = np.random.default_rng(len(name))
    n = 257
    times = rng.uniform(-0.1, clip.duration + 0.1, size=n).astype(np.float32)
    oracle = ob.oracle_decompress_tracks_batch([clip.blob], np.zeros(n, dtype=np.uint32), times, clip.num_tracks)
    handles = np.full(n, handle, dtype=np.uint32)
    for skip in ((0, 0, 0), (0, 0, 1), (1, 0, 0), (0, 1, 1)):
        got = launch(context, handles, times, layout, skip, max_tracks=clip.num_tracks)
        assert helpers.exact(got, expected_through_layout(oracle, layout, skip)), (name, layout, skip)
    context.unregister_clip(handle)
    assert context.rejected_instance_count() == 0


def test_layouts_with_rows_and_mixed_clips(context):
    """several clips of different sizes in one batch, poses scattered to rows of a larger buffer: tracks beyond a smaller clip's
    count and rows nobody writes keep the caller's bytes"""
    rng = np.random.default_rng(5)
    clips = [synth.build_clip(seed=900 + i, num_tracks=tracks, num_samples=int(rng.integers(2, 80)), has_scale=i % 2) for i, tracks in enumerate([3, 40, 107, 130])]
    handles = np.array([context.register_clip(c.blob) for c in clips], dtype=np.uint32)
    max_tracks = 130
    n, num_rows = 600, 1000
    which = rng.integers(0, len(clips), size=n)
    times = np.array([rng.uniform(0.0, clips[c].duration) for c in which], dtype=np.float32)
    rows = rng.choice(num_rows, size=n, replace=False).astype(np.uint32)
    oracle = ob.oracle_decompress_tracks_batch([c.blob for c in clips], which, times, max_tracks)
    num_tracks = np.array([c.num_tracks for c in clips])[which]
    for layout in ("qvv40", "qv32", "qvv48"):
        skip = (0, 0, 0) if layout != "qvv48" else (0, 1, 0)
        got = launch(context, handles[which], times, layout, skip, rows=rows, num_rows=num_rows, max_tracks=max_tracks)
        expected = np.full_like(got, FILL)
        through = expected_through_layout(oracle, layout, skip)
        for i in range(n):
            expected[rows[i], : num_tracks[i]] = through[i, : num_tracks[i]]
        assert helpers.exact(got, expected), layout
    for handle in handles:
        context.unregister_clip(int(handle))


def test_default_modes_and_per_track_rounding_through_a_layout(context):
    """the any-settings paths (skipped / caller supplied defaults, per track rounding) store through the layout as well"""
    clip = synth.build_clip(**CLIP_SPECS["scale_37"])
    handle = context.register_clip(clip.blob)
    rng = np.random.default_rng(9)
    n = 130
    times = rng.uniform(0.0, clip.duration, size=n).astype(np.float32)
    handles = np.full(n, handle, dtype=np.uint32)

    skipped = runtime.default_params(default_rotation_mode=runtime.DEFAULT_SKIPPED, default_translation_mode=runtime.DEFAULT_SKIPPED, default_scale_mode=runtime.DEFAULT_SKIPPED)
    options = ob.default_options(default_rotation_mode=ob.DEFAULT_SKIPPED, default_translation_mode=ob.DEFAULT_SKIPPED, default_scale_mode=ob.DEFAULT_SKIPPED)
    oracle = ob.oracle_decompress_tracks_batch([clip.blob], np.zeros(n, dtype=np.uint32), times, clip.num_tracks, options=options,
                                               out=np.full((n, clip.num_tracks, 12), np.nan, dtype=np.float32))
    written = ~np.isnan(oracle)
    for layout in ("qvv40", "qv32"):
        got = launch(context, handles, times, layout, max_tracks=clip.num_tracks, params=skipped)
        expected = expected_through_layout(np.where(written, oracle, FILL).astype(np.float32), layout, (0, 0, 0))
        assert helpers.exact(got, expected), layout

    d_policies = torch.from_numpy(rng.integers(0, 4, size=clip.num_tracks).astype(np.uint8)).cuda()
    per_track = runtime.default_params(rounding_policy=runtime.ROUND_PER_TRACK, per_track_rounding=1)
    per_track.track_rounding_policies = d_policies.data_ptr()
    policies = d_policies.cpu().numpy()
    options = ob.default_options(per_track_rounding=1)
    options.track_rounding = policies.ctypes.data
    oracle = ob.oracle_decompress_tracks_batch([clip.blob], np.zeros(n, dtype=np.uint32), times, clip.num_tracks, rounding=ob.ROUND_PER_TRACK, options=options)
    got = launch(context, handles, times, "qvv40", max_tracks=clip.num_tracks, params=per_track)
    assert helpers.exact(got, expected_through_layout(oracle, "qvv40", (0, 0, 0)))
    context.unregister_clip(handle)


def test_unknown_layout_is_refused(context):
    clip = synth.build_clip(**CLIP_SPECS["two_samples_three_tracks"])
    handle = context.register_clip(clip.blob)
    output = runtime.OutputDesc()
    output.layout = 9
    d = torch.zeros(64, dtype=torch.float32, device="cuda")
    d_handles = torch.full((1,), handle, dtype=torch.int32, device="cuda")
    with pytest.raises(runtime.AclHipError):
        context.decompress_tracks_batch_out(d_handles.data_ptr(), d.data_ptr(), 1, d.data_ptr(), 144, output)
    context.unregister_clip(handle)


def test_host_entry_point_and_cpp_writer_switches():
    """aclhip_decompress_tracks_host_out: the path the C++ mirror takes for a writer with skip_all_* switches"""
    import ctypes
    with runtime.Context(0) as context:
        clip = synth.build_clip(**CLIP_SPECS["scale_37"])
        handle = context.register_clip(clip.blob)
        rng = np.random.default_rng(3)
        n = 19
        times = rng.uniform(0.0, clip.duration, size=n).astype(np.float32)
        handles = np.full(n, handle, dtype=np.uint32)
        oracle = ob.oracle_decompress_tracks_batch([clip.blob], np.zeros(n, dtype=np.uint32), times, clip.num_tracks)
        lib = runtime.load_library()
        params = runtime.default_params()
        for layout, skip in (("qvv48", (0, 1, 0)), ("qvv40", (0, 0, 0)), ("qv32", (1, 0, 0))):
            layout_id, bytes_per_track = runtime.LAYOUTS[layout]
            out = np.full((n, clip.num_tracks, bytes_per_track // 4), FILL, dtype=np.float32)
            output = runtime.OutputDesc()
            output.layout = layout_id
            output.skip_rotations, output.skip_translations, output.skip_scales = skip
            status = lib.aclhip_decompress_tracks_host_out(context._handle, handles.ctypes.data, times.ctypes.data, n, ctypes.byref(params), 0, ctypes.byref(output),
                                                           out.ctypes.data, clip.num_tracks * bytes_per_track)
            assert status == 0
            assert helpers.exact(out, expected_through_layout(oracle, layout, skip)), layout
        context.unregister_clip(handle)


@pytest.mark.parametrize("name", ["cmu_100", "cinematic_300", "raw_and_constant_rates"])
@pytest.mark.parametrize("layout", ["qvv48", "qvv40", "qv32"])
def test_per_track_skips_leave_the_callers_bytes(context, name, layout):
    """track_writer::skip_track_rotation / _translation / _scale(track_index) (core/track_writer.h:189-191) for a whole launch:
    aclhip_output_desc::skip_tracks, one byte per track. Oracle values where nothing is masked, the caller's bytes where a sub-track
    is -- on top of the kinds skipped altogether, in every layout, through the device and the host entry points."""
    clip = synth.build_clip(**CLIP_SPECS[name])
    handle = context.register_clip(clip.blob)
    rng = np.random.default_rng(len(name) + len(layout))
    n = 130
    times = rng.uniform(0.0, clip.duration, size=n).astype(np.float32)
    oracle = ob.oracle_decompress_tracks_batch([clip.blob], np.zeros(n, dtype=np.uint32), times, clip.num_tracks)
    handles = np.full(n, handle, dtype=np.uint32)
    skip_tracks = rng.integers(0, 8, size=clip.num_tracks).astype(np.uint8)
    skip_tracks[rng.uniform(size=clip.num_tracks) < 0.5] = 0          # half of the tracks keep everything
    layout_id, bytes_per_track = runtime.LAYOUTS[layout]
    width = bytes_per_track // 4
    for skip in ((0, 0, 0), (0, 1, 0)):
        got = launch(context, handles, times, layout, skip, max_tracks=clip.num_tracks, skip_tracks=skip_tracks)
        expected = expected_through_layout(oracle, layout, skip)
        # lanes of the masked sub-tracks go back to the fill value
        lanes = {"qvv48": ((0, 4), (4, 8), (8, 12)), "qvv40": ((0, 4), (4, 7), (7, 10)), "qv32": ((0, 4), (4, 8), None)}[layout]
        for kind in range(3):
            if lanes[kind] is None:
                continue
            masked = (skip_tracks >> kind) & 1 == 1
            expected[:, masked, lanes[kind][0]: lanes[kind][1]] = FILL
        assert helpers.exact(got, expected), (name, layout, skip)
    # the host convenience entry point takes the mask as a host array
    out = np.full((n, clip.num_tracks, width), FILL, dtype=np.float32)
    output = runtime.OutputDesc()
    output.layout = layout_id
    mask_host = np.ascontiguousarray(skip_tracks)
    output.skip_tracks = mask_host.ctypes.data
    params = runtime.default_params()
    context._check(context._lib.aclhip_decompress_tracks_host_out(context._handle, handles.ctypes.data, times.ctypes.data, n, __import__("ctypes").byref(params), 0,
                                                                  __import__("ctypes").byref(output), out.ctypes.data, clip.num_tracks * bytes_per_track))
    expected = expected_through_layout(oracle, layout, (0, 0, 0))
    for kind in range(3):
        if lanes[kind] is not None:
            expected[:, (skip_tracks >> kind) & 1 == 1, lanes[kind][0]: lanes[kind][1]] = FILL
    assert helpers.exact(out, expected)
    context.unregister_clip(handle)
