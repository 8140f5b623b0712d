"""A launch is shaped by the batch it decodes and checked against every clip it meets (round 4; the reference sizes its work per clip,
includes/acl/decompression/impl/decompression.transform.h:1526-1540, and returns silently on misuse, :1532-1537):
  * waves per instance and LDS per wave follow the pose stride (what a row can hold), not the largest clip of the registry;
  * a captured hipGraph replayed after a LARGER clip was registered refuses that clip's instances (counted, rows untouched, the
    neighbours bit exact) instead of decoding them into too few windows or into the next wave's LDS slot;
  * a pose stride smaller than a clip's pose refuses the instance instead of writing into the next row or past the buffer.
Through the C ABI, every pose compared with the CPU oracle bit for bit. Needs a GPU."""
import numpy as np
import pytest

from acl_amd import runtime, synth
from oracle import bindings as ob
import helpers

pytestmark = pytest.mark.gpu

SENTINEL = np.float32(-12345.5)


@pytest.fixture()
def setup():
    import torch
    device = torch.device("cuda:0")
    ctx = runtime.Context(0)
    yield ctx, torch, device
    ctx.close()


def _clip(seed, num_tracks, **kwargs):
    return synth.build_clip(seed=seed, num_tracks=num_tracks, num_samples=kwargs.pop("num_samples", 61), sample_rate=30.0, **kwargs)


def _oracle(clips, which, times):
    """per instance oracle poses (list of [num_tracks, 12])"""
    return [ob.oracle_decompress_tracks(clips[c].blob, float(t)) for c, t in zip(which, times)]


def test_the_launch_follows_the_pose_stride_not_the_registry(setup):
    ctx, torch, device = setup
    character, rig, leader = _clip(11, 100), _clip(12, 300, has_scale=1, scale_default=0.4), _clip(13, 551, num_samples=20)
    handles = [ctx.register_clip(c.blob) for c in (character, rig, leader)]
    # 104 tracks per pose window
    assert ctx.pose_windows_of_launch(100 * 48) == 1
    assert ctx.pose_windows_of_launch(104 * 48) == 1
    assert ctx.pose_windows_of_launch(105 * 48) == 2
    assert ctx.pose_windows_of_launch(300 * 48) == 3
    assert ctx.pose_windows_of_launch(551 * 48) == 6
    assert ctx.pose_windows_of_launch(4000 * 48) == 6                       # never more than the largest registered clip needs
    assert ctx.pose_windows_of_launch(100 * 32, layout=runtime.LAYOUT_QV32) == 1
    assert ctx.pose_windows_of_launch(100 * 48, layout=runtime.LAYOUT_QV32) == 2   # 150 tracks of 32 bytes
    import os
    knobs = any(name in os.environ for name in ("ACLHIP_IN_TURN_ITEMS", "ACLHIP_IN_TURN_ADJACENT", "ACLHIP_WIDE_KEY_LOADS", "ACLHIP_FORCE_GENERIC_KERNEL"))
    assert knobs or ctx.tracks_kernel_name(pose_stride_bytes=4800) == "decompress_tracks_kernel"
    assert knobs or ctx.tracks_kernel_name(pose_stride_bytes=14400) == "decompress_tracks_in_turn_kernel"          # several windows: items in turn, 16 byte key reads
    assert knobs or ctx.tracks_kernel_name() == "decompress_tracks_in_turn_kernel"       # rows as wide as the registry's largest clip

    n = 512
    rng = np.random.default_rng(5)
    clips = (character, rig, leader)
    for which_clip, stride in ((0, 4800), (1, 14400), (2, 551 * 48), (0, 551 * 48)):
        clip = clips[which_clip]
        times = rng.uniform(0.0, clip.duration, size=n).astype(np.float32)
        d_clips = torch.full((n,), handles[which_clip], dtype=torch.int32, device=device)
        d_times = torch.from_numpy(times).to(device)
        d_poses = torch.full((n, stride // 4), float(SENTINEL), dtype=torch.float32, device=device)
        ctx.decompress_tracks_batch(d_clips.data_ptr(), d_times.data_ptr(), n, d_poses.data_ptr(), stride)
        torch.cuda.synchronize(device)
        poses = d_poses.cpu().numpy()
        expected = ob.oracle_decompress_tracks_batch([clip.blob], np.zeros(n, dtype=np.uint32), times, clip.num_tracks)
        assert helpers.exact(poses[:, : clip.num_tracks * 12].reshape(n, clip.num_tracks, 12), expected), (which_clip, stride)
        assert np.all(poses[:, clip.num_tracks * 12:] == SENTINEL)
    assert ctx.rejected_instance_count() == 0


def test_a_captured_graph_refuses_clips_registered_after_its_capture(setup):
    """a captured launch holds the address of the clip table, which never moves -- so a replay can meet a clip that did
    not exist when the launch was shaped. 70 bones at capture: one window, 256 quads of LDS per wave. Afterwards a 100-bone clip (300
    quads: too large for the LDS slot) and a 300-bone rig (three windows: the launch has one wave per instance) are registered and the
    replayed instance list names them."""
    ctx, torch, device = setup
    small, medium, large = _clip(21, 70), _clip(22, 100), _clip(23, 300, has_scale=1)
    h_small = ctx.register_clip(small.blob)
    n, stride = 1024, 300 * 48
    rng = np.random.default_rng(9)
    times = rng.uniform(0.0, small.duration, size=n).astype(np.float32)
    d_clips = torch.full((n,), h_small, dtype=torch.int32, device=device)
    d_times = torch.from_numpy(times).to(device)
    d_poses = torch.full((n, stride // 4), float(SENTINEL), dtype=torch.float32, device=device)
    assert ctx.pose_windows_of_launch(stride) == 1

    graph = torch.cuda.CUDAGraph()
    capture_stream = torch.cuda.Stream(device)
    torch.cuda.synchronize(device)
    with torch.cuda.graph(graph, stream=capture_stream):
        ctx.decompress_tracks_batch(d_clips.data_ptr(), d_times.data_ptr(), n, d_poses.data_ptr(), stride, stream=torch.cuda.current_stream(device).cuda_stream)
    graph.replay()
    torch.cuda.synchronize(device)
    expected_small = ob.oracle_decompress_tracks_batch([small.blob], np.zeros(n, dtype=np.uint32), times, 70)
    assert helpers.exact(d_poses.cpu().numpy()[:, : 70 * 12].reshape(n, 70, 12), expected_small)
    assert ctx.rejected_instance_count() == 0

    h_medium, h_large = ctx.register_clip(medium.blob), ctx.register_clip(large.blob)
    which = rng.integers(0, 3, size=n)
    which[:8] = [0, 1, 2, 0, 2, 1, 0, 0]
    handles = np.array([h_small, h_medium, h_large], dtype=np.int32)
    d_clips.copy_(torch.from_numpy(handles[which]))
    d_poses.fill_(float(SENTINEL))
    graph.replay()
    torch.cuda.synchronize(device)
    poses = d_poses.cpu().numpy()
    refused = which != 0
    assert ctx.rejected_instance_count() == int(refused.sum())
    assert np.all(poses[refused] == SENTINEL)                                   # nothing of a refused instance was written
    assert helpers.exact(poses[~refused][:, : 70 * 12].reshape(-1, 70, 12), expected_small[~refused])
    assert np.all(poses[~refused][:, 70 * 12:] == SENTINEL)

    # the same arguments launched NOW are shaped for what the registry holds now: every instance decodes
    d_poses.fill_(float(SENTINEL))
    ctx.decompress_tracks_batch(d_clips.data_ptr(), d_times.data_ptr(), n, d_poses.data_ptr(), stride)
    torch.cuda.synchronize(device)
    poses = d_poses.cpu().numpy()
    clips = (small, medium, large)
    for i, pose in enumerate(_oracle(clips, which[:64], times[:64])):
        tracks = clips[which[i]].num_tracks
        assert helpers.exact(poses[i, : tracks * 12].reshape(tracks, 12), pose), i
    assert ctx.rejected_instance_count() == int(refused.sum())


@pytest.mark.parametrize("layout", ["qvv48", "qvv40", "qv32"])
@pytest.mark.parametrize("any_settings", [False, True])
def test_a_stride_too_small_for_a_clip_refuses_the_instance(setup, layout, any_settings):
    """70-bone and 100-bone instances in rows that hold 70 bones: the 100-bone ones are refused and counted, no byte outside a row's
    own pose is written -- not in the next row, not behind the buffer."""
    ctx, torch, device = setup
    layout_id, bytes_per_track = runtime.LAYOUTS[layout]
    small, medium = _clip(31, 70), _clip(32, 100)
    handles = np.array([ctx.register_clip(small.blob), ctx.register_clip(medium.blob)], dtype=np.int32)
    n = 640
    stride = (70 * bytes_per_track + 63) // 64 * 64
    rng = np.random.default_rng(3)
    which = rng.integers(0, 2, size=n)
    which[-1] = 1                                                              # the last row: an overrun would leave the buffer
    times = rng.uniform(0.0, small.duration, size=n).astype(np.float32)
    d_clips = torch.from_numpy(handles[which]).to(device)
    d_times = torch.from_numpy(times).to(device)
    guard_rows = 4
    d_poses = torch.full((n + guard_rows, stride // 4), float(SENTINEL), dtype=torch.float32, device=device)
    params = runtime.default_params(normalization=runtime.NORMALIZE_ALWAYS) if any_settings else runtime.default_params()
    output = runtime.OutputDesc()
    output.layout = layout_id
    ctx.decompress_tracks_batch_out(d_clips.data_ptr(), d_times.data_ptr(), n, d_poses.data_ptr(), stride, output, params=params)
    torch.cuda.synchronize(device)
    poses = d_poses.cpu().numpy()
    refused = which == 1
    assert ctx.rejected_instance_count() == int(refused.sum())
    assert np.all(poses[n:] == SENTINEL)
    assert np.all(poses[:n][refused] == SENTINEL)
    options = ob.default_options(normalization=ob.NORMALIZE_ALWAYS) if any_settings else None
    floats_per_track = bytes_per_track // 4
    for i in np.flatnonzero(~refused)[:48]:
        expected = ob.oracle_decompress_tracks(small.blob, float(times[i]), 0, options)
        expected_row = runtime.relayout_pose(expected, layout_id)
        assert helpers.exact(poses[i, : 70 * floats_per_track], np.asarray(expected_row, dtype=np.float32).reshape(-1)[: 70 * floats_per_track]), (layout, i)
        assert np.all(poses[i, 70 * floats_per_track:] == SENTINEL)


def test_pose_consumers_are_shaped_by_the_batch(setup):
    """One 3 600-bone asset in the registry (too large for the consumers' LDS images) used to switch object space off for every clip;
    now the 100-bone batch in 4 800 byte rows decodes, and an instance of the asset in such rows is refused."""
    ctx, torch, device = setup
    character = _clip(41, 100)
    asset = synth.build_clip(seed=42, num_tracks=3600, num_samples=4, sample_rate=30.0, has_scale=1, scale_default=0.5)       # (with scale: 48 bytes per transform in LDS)
    h_character, h_asset = ctx.register_clip(character.blob), ctx.register_clip(asset.blob)
    parents = synth.humanoid_hierarchy(100)
    ctx.set_clip_hierarchy(h_character, parents)
    ctx.set_clip_hierarchy(h_asset, np.concatenate([[runtime.NO_PARENT], np.arange(3599, dtype=np.uint32) // 2]).astype(np.uint32))
    n = 256
    rng = np.random.default_rng(17)
    times = rng.uniform(0.0, character.duration, size=n).astype(np.float32)
    which = np.zeros(n, dtype=np.int64)
    which[[3, 77, 255]] = 1
    handles = np.array([h_character, h_asset], dtype=np.int32)
    d_clips = torch.from_numpy(handles[which]).to(device)
    d_times = torch.from_numpy(times).to(device)
    d_poses = torch.full((n + 2, 1200), float(SENTINEL), dtype=torch.float32, device=device)
    consumers = runtime.PoseConsumers()
    consumers.object_space = 1
    ctx.decompress_poses_batch(d_clips.data_ptr(), d_times.data_ptr(), n, d_poses.data_ptr(), 4800, consumers)
    torch.cuda.synchronize(device)
    poses = d_poses.cpu().numpy()
    assert ctx.rejected_instance_count() == 3
    assert np.all(poses[n:] == SENTINEL) and np.all(poses[:n][which == 1] == SENTINEL)
    expected = ob.oracle_decompress_poses_batch([character.blob], np.zeros(n, dtype=np.uint32), times, 100, parent_indices=parents)
    keep = which == 0
    assert helpers.bit_equal(poses[:n][keep].reshape(-1, 100, 12), expected[keep])
    # rows wide enough for the asset ask for LDS images the device does not have: the call says so
    with pytest.raises(runtime.AclHipError):
        ctx.decompress_poses_batch(d_clips.data_ptr(), d_times.data_ptr(), n, d_poses.data_ptr(), 3600 * 48, consumers)


def test_a_captured_object_space_launch_follows_a_longer_walk_schedule_from_global_memory(setup):
    """The walk schedule a workgroup keeps in LDS is sized when the launch is enqueued; a hierarchy replaced behind a captured launch's
    back by one with a longer schedule (a chain: one step per transform) is walked from global memory instead."""
    ctx, torch, device = setup
    clip = _clip(51, 60)
    handle = ctx.register_clip(clip.blob)
    flat = np.zeros(60, dtype=np.uint32)
    flat[0] = runtime.NO_PARENT
    chain = np.concatenate([[runtime.NO_PARENT], np.arange(59, dtype=np.uint32)]).astype(np.uint32)
    ctx.set_clip_hierarchy(handle, flat)
    n = 512
    rng = np.random.default_rng(23)
    times = rng.uniform(0.0, clip.duration, size=n).astype(np.float32)
    d_clips = torch.full((n,), handle, dtype=torch.int32, device=device)
    d_times = torch.from_numpy(times).to(device)
    d_poses = torch.zeros((n, 60, 12), dtype=torch.float32, device=device)
    consumers = runtime.PoseConsumers()
    consumers.object_space = 1
    graph = torch.cuda.CUDAGraph()
    capture_stream = torch.cuda.Stream(device)
    torch.cuda.synchronize(device)
    with torch.cuda.graph(graph, stream=capture_stream):
        ctx.decompress_poses_batch(d_clips.data_ptr(), d_times.data_ptr(), n, d_poses.data_ptr(), 60 * 48, consumers, stream=torch.cuda.current_stream(device).cuda_stream)
    graph.replay()
    torch.cuda.synchronize(device)
    assert helpers.bit_equal(d_poses.cpu().numpy(), ob.oracle_decompress_poses_batch([clip.blob], np.zeros(n, dtype=np.uint32), times, 60, parent_indices=flat))
    ctx.set_clip_hierarchy(handle, chain)
    graph.replay()
    torch.cuda.synchronize(device)
    assert helpers.bit_equal(d_poses.cpu().numpy(), ob.oracle_decompress_poses_batch([clip.blob], np.zeros(n, dtype=np.uint32), times, 60, parent_indices=chain))
    assert ctx.rejected_instance_count() == 0


def test_scalar_launches_are_shaped_by_the_batch_and_checked(setup):
    ctx, torch, device = setup
    curves = synth.build_scalar_clip(seed=61, track_type=0, num_tracks=64, num_samples=90)
    many = synth.build_scalar_clip(seed=62, track_type=0, num_tracks=700, num_samples=9)
    wide = synth.build_scalar_clip(seed=63, track_type=4, num_tracks=64, num_samples=30)
    h_curves, h_many, h_wide = ctx.register_clip(curves.blob), ctx.register_clip(many.blob), ctx.register_clip(wide.blob)
    n = 20000                                                                   # (the grouped kernel: batches of 16 384 and more)
    rng = np.random.default_rng(29)
    which = np.zeros(n, dtype=np.int64)
    which[rng.choice(n, size=40, replace=False)] = rng.integers(1, 3, size=40)
    which[-1] = 1
    times = rng.uniform(0.0, curves.duration, size=n).astype(np.float32)
    handles = np.array([h_curves, h_many, h_wide], dtype=np.int32)
    d_clips = torch.from_numpy(handles[which]).to(device)
    d_times = torch.from_numpy(times).to(device)
    for count in (n, 600):                                                      # grouped and per instance kernels
        d_values = torch.full((count + 8, 64), float(SENTINEL), dtype=torch.float32, device=device)
        before = ctx.rejected_instance_count()
        ctx.decompress_scalar_tracks_batch(d_clips.data_ptr() + (n - count) * 4, d_times.data_ptr() + (n - count) * 4, count, d_values.data_ptr(), 256)
        torch.cuda.synchronize(device)
        values = d_values.cpu().numpy()
        tail = which[n - count:]
        assert ctx.rejected_instance_count() - before == int((tail != 0).sum())   # 700 curves / 64 x vector4f do not fit 256 byte rows
        assert np.all(values[count:] == SENTINEL) and np.all(values[:count][tail != 0] == SENTINEL)
        expected = ob.oracle_scalar_decompress_tracks_batch([curves.blob], np.zeros(count, dtype=np.uint32), times[n - count:], 64)
        assert helpers.exact(values[:count][tail == 0], expected[tail == 0])
