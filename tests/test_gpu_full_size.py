"""BASELINE.json-sized batches on the GPU through the device-pointer C ABI (torch only provides device memory and streams):
EVERY instance of every batch is compared with the oracle bit for bit (the reference's validator checks every sample too,
tools/acl_compressor/sources/validate_tracks.cpp:92-260), plus the size independent properties that validator relies on
(:170-258). Shapes follow BASELINE.json configs[1..4] as SURVEY 8(d) writes them."""
import numpy as np
import pytest

from acl_amd import runtime, synth
from oracle import bindings as ob
import helpers

pytestmark = pytest.mark.gpu

N = 65536


@pytest.fixture(scope="module")
def setup():
    import torch
    ctx = runtime.Context(0)
    device = torch.device("cuda:0")
    yield ctx, torch, device
    ctx.close()


def _decode(ctx, torch, device, handles, which, times, max_tracks, params=None, stream=None):
    d_clips = torch.from_numpy(np.asarray(handles, dtype=np.uint32)[which].astype(np.int32)).to(device)
    d_times = torch.from_numpy(times).to(device)
    d_poses = torch.zeros((times.size, max_tracks, 12), dtype=torch.float32, device=device)
    stream = stream or torch.cuda.current_stream(device)
    stream.wait_stream(torch.cuda.current_stream(device))       # the uploads and the zero fill above ran on the current stream (side streams do not wait for it on their own)
    ctx.decompress_tracks_batch(d_clips.data_ptr(), d_times.data_ptr(), times.size, d_poses.data_ptr(), max_tracks * 48, params=params, stream=stream.cuda_stream)
    stream.synchronize()
    return d_poses


def test_64k_instances_of_one_100_bone_clip(setup):
    ctx, torch, device = setup
    clip = synth.build_clip(seed=2, num_tracks=100, num_samples=301, sample_rate=30.0)      # BASELINE.json configs[1]
    handle = ctx.register_clip(clip.blob)
    rng = np.random.default_rng(21)
    times = rng.uniform(0.0, clip.duration, size=N).astype(np.float32)
    d_poses = _decode(ctx, torch, device, [handle], np.zeros(N, dtype=np.int64), times, 100)
    poses = d_poses.cpu().numpy()

    # every instance against the oracle, bit for bit
    expected = ob.oracle_decompress_tracks_batch([clip.blob], np.zeros(N, dtype=np.uint32), times, 100)
    assert np.array_equal(poses.view(np.uint32), expected.view(np.uint32))

    # every rotation is a unit quaternion with finite components, every vector finite
    assert np.isfinite(poses).all()
    assert np.abs(np.linalg.norm(poses[:, :, :4], axis=2) - 1.0).max() <= 1e-5

    # idempotence: decoding the same batch again gives identical bits
    again = _decode(ctx, torch, device, [handle], np.zeros(N, dtype=np.int64), times, 100).cpu().numpy()
    assert np.array_equal(poses.view(np.uint32), again.view(np.uint32))

    # a second HIP stream gives the same result (stream ordered API)
    side_stream = torch.cuda.Stream(device)
    other = _decode(ctx, torch, device, [handle], np.zeros(N, dtype=np.int64), times, 100, stream=side_stream).cpu().numpy()
    assert np.array_equal(poses.view(np.uint32), other.view(np.uint32))

    # clamping: seek(-0.2) == seek(0), seek(duration + 1) == seek(duration)  (validate_tracks.cpp:170-187)
    edge_times = np.array([-0.2, 0.0, clip.duration + 1.0, clip.duration], dtype=np.float32)
    edges = _decode(ctx, torch, device, [handle], np.zeros(4, dtype=np.int64), edge_times, 100).cpu().numpy()
    assert np.array_equal(edges[0].view(np.uint32), edges[1].view(np.uint32))
    assert np.array_equal(edges[2].view(np.uint32), edges[3].view(np.uint32))

    # checksum of checksums: instances with equal sample times produce equal poses wherever they sit in the batch
    repeated = np.tile(times[:1024], N // 1024)
    tiled = _decode(ctx, torch, device, [handle], np.zeros(N, dtype=np.int64), repeated, 100).cpu().numpy().view(np.uint32)
    assert (tiled.reshape(N // 1024, 1024, -1) == tiled[:1024].reshape(1, 1024, -1)).all()
    ctx.unregister_clip(handle)


def test_64k_instances_from_256_distinct_clips(setup):
    ctx, torch, device = setup
    spec_rng = np.random.default_rng(3)                                                       # BASELINE.json configs[2]
    clips = []
    for i in range(256):
        animated = spec_rng.uniform(0.25, 0.5)
        clips.append(synth.build_clip(seed=300 + i, num_tracks=100, num_samples=int(spec_rng.integers(31, 601)), sample_rate=30.0,
                                      rotation_default=0.02, rotation_constant=float(0.98 - animated),
                                      wrap=int(spec_rng.uniform() < 0.1), strip_keyframes=int(spec_rng.uniform() < 0.1),
                                      min_bits=int(spec_rng.integers(5, 10)), max_bits=int(spec_rng.integers(12, 19))))
    handles = [ctx.register_clip(c.blob) for c in clips]
    rng = np.random.default_rng(22)
    which = rng.integers(0, 256, size=N)
    durations = np.array([c.duration for c in clips], dtype=np.float32)
    times = (rng.uniform(0, 1, size=N).astype(np.float32) * durations[which]).astype(np.float32)
    poses = _decode(ctx, torch, device, handles, which, times, 100).cpu().numpy()
    expected = ob.oracle_decompress_tracks_batch([c.blob for c in clips], which, times, 100)
    assert np.array_equal(poses.view(np.uint32), expected.view(np.uint32))

    # bucketing the instance list by clip only permutes the output
    order = np.argsort(which, kind="stable")
    sorted_poses = _decode(ctx, torch, device, handles, which[order], times[order], 100).cpu().numpy()
    assert np.array_equal(sorted_poses.view(np.uint32), poses[order].view(np.uint32))
    assert np.abs(np.linalg.norm(poses[:, :, :4], axis=2) - 1.0).max() <= 1e-5
    assert ctx.rejected_instance_count() == 0
    for h in handles:
        ctx.unregister_clip(h)


def test_300_bone_rig_with_scale(setup):
    ctx, torch, device = setup
    clip = synth.build_clip(seed=4, num_tracks=300, num_samples=451, sample_rate=30.0, has_scale=1,          # configs[3] shard shape
                            scale_default=0.7, scale_constant=0.1, rotation_constant=0.45, translation_constant=0.8)
    handle = ctx.register_clip(clip.blob)
    rng = np.random.default_rng(23)
    # the whole 65 536 instance shard of configs[3], every instance against the oracle (in pieces: 944 MB of poses)
    all_times = rng.uniform(0.0, clip.duration, size=N).astype(np.float32)
    d_all = _decode(ctx, torch, device, [handle], np.zeros(N, dtype=np.int64), all_times, 300)
    piece = 8192
    for first in range(0, N, piece):
        got = d_all[first: first + piece].cpu().numpy()
        expected = ob.oracle_decompress_tracks_batch([clip.blob], np.zeros(piece, dtype=np.uint32), all_times[first: first + piece], 300)
        assert np.array_equal(got.view(np.uint32), expected.view(np.uint32)), f"instances [{first}, {first + piece})"
    n = 16384
    times = all_times[:n]
    poses = d_all[:n].cpu().numpy()
    del d_all

    # global rounding policy == per track rounding policy with the same value on every track (validate_tracks.cpp:189-211)
    d_policies = torch.full((300,), runtime.ROUND_CEIL, dtype=torch.uint8, device=device)
    global_params = runtime.default_params(rounding_policy=runtime.ROUND_CEIL)
    per_track_params = runtime.default_params(rounding_policy=runtime.ROUND_PER_TRACK, per_track_rounding=1)
    per_track_params.track_rounding_policies = d_policies.data_ptr()
    a = _decode(ctx, torch, device, [handle], np.zeros(n, dtype=np.int64), times, 300, params=global_params).cpu().numpy()
    b = _decode(ctx, torch, device, [handle], np.zeros(n, dtype=np.int64), times, 300, params=per_track_params).cpu().numpy()
    # translations/scales exactly; rotations up to sign and up to the normalization the global path applies after its lerp
    # with alpha = 1 (the per track path returns the raw keyframe, whose length is off by the quantization error)
    assert np.array_equal(a[:, :, 4:].view(np.uint32), b[:, :, 4:].view(np.uint32))
    b_normalized = b[:, :, :4] / np.linalg.norm(b[:, :, :4], axis=2, keepdims=True)
    assert np.abs(np.abs(a[:, :, :4]) - np.abs(b_normalized)).max() <= 1e-6

    # the three default modes agree on non default sub-tracks (validate_tracks.cpp:220-229)
    skipped = runtime.default_params(default_rotation_mode=0, default_translation_mode=0, default_scale_mode=0)
    c = _decode(ctx, torch, device, [handle], np.zeros(n, dtype=np.int64), times, 300, params=skipped).cpu().numpy()
    reference_pose = poses
    written = c != 0.0
    assert np.array_equal(c[written].view(np.uint32), reference_pose[written].view(np.uint32))
    ctx.unregister_clip(handle)


def test_database_tiers_streamed_in_while_batches_run(setup):
    """configs[4] as SURVEY 8(d)5 writes it: 64 clips bound to one database (medium 0 % / low 50 %, the reference's default tier
    proportions), the 32 768 instance per-GPU shard of 262 144 instances over 8 GPUs; the low importance tier arrives (and partly
    leaves again) between batches on the decode stream. EVERY instance of every batch is compared with the oracle in the same
    streaming state (oracle/database.py restates database_context)."""
    from oracle.database import OracleDatabase
    ctx, torch, device = setup
    case = helpers.load_bench_database()
    database = ctx.register_database(case["database"], case["bulk_medium"] if case["bulk_medium"].size else None, case["bulk_low"] if case["bulk_low"].size else None)
    handles = [ctx.register_clip_with_database(clip, database) for clip in case["clips"]]
    assert len(handles) == 64
    oracle_db = OracleDatabase(case["database"], case["bulk_medium"], case["bulk_low"])
    max_tracks = max(ob.oracle().aclo_num_tracks(clip.ctypes.data) for clip in case["clips"])
    durations = np.array([ob.oracle().aclo_finite_duration(clip.ctypes.data, ob.LOOP_AS_COMPRESSED) for clip in case["clips"]], dtype=np.float32)

    rng = np.random.default_rng(31)
    n = 32768
    which = rng.integers(0, len(handles), size=n)
    times = (rng.uniform(0.0, 1.0, size=n).astype(np.float32) * durations[which]).astype(np.float32)
    stream = torch.cuda.Stream(device)

    # (stream_in?, tier, num_chunks): chunk by chunk, a partial stream_out that leaves a hole (the reference does not refill holes,
    # database.impl.h:478-497), the rest, requests with nothing left to do, everything out and back in
    schedule = [None, (True, 2, 1), (True, 2, 1), (True, 2, 3), (True, 2, 5), (False, 2, 4), (True, 2, 2), (True, 2, 0xFFFFFFFF),
                (True, 1, 0xFFFFFFFF), (True, 2, 0xFFFFFFFF), (False, 2, 0xFFFFFFFF), (True, 2, 0xFFFFFFFF)]
    previous = None
    for request in schedule:
        moved = 0
        if request is not None:
            stream_in, tier, num_chunks = request
            if stream_in:
                moved = ctx.database_stream_in(database, tier, num_chunks, stream=stream.cuda_stream)
                assert moved == oracle_db.stream_in(tier, num_chunks)
            else:
                moved = ctx.database_stream_out(database, tier, num_chunks, stream=stream.cuda_stream)
                assert moved == oracle_db.stream_out(tier, num_chunks)
        poses = _decode(ctx, torch, device, handles, which, times, max_tracks, stream=stream).cpu().numpy()
        expected = ob.oracle_decompress_tracks_batch(case["clips"], which, times, max_tracks, options=oracle_db.options())
        assert helpers.bit_equal(poses, expected), f"after request {request}"
        if previous is not None and moved != 0:
            assert not np.array_equal(previous, poses)          # keyframes arriving or leaving changed some poses
        previous = poses
    assert oracle_db.is_streamed_in(2)

    # everything resident: sample times that fall on a keyframe decode to the same pose whatever the rounding policy
    sample_times = (np.floor(times * 30.0) / 30.0).astype(np.float32)
    a = _decode(ctx, torch, device, handles, which, sample_times, max_tracks, params=runtime.default_params(rounding_policy=runtime.ROUND_FLOOR), stream=stream).cpu().numpy()
    b = _decode(ctx, torch, device, handles, which, sample_times, max_tracks, params=runtime.default_params(rounding_policy=runtime.ROUND_NONE), stream=stream).cpu().numpy()
    sample_indices = sample_times * np.float32(30.0)          # the kernel's own fp32 product
    exact = sample_indices == np.round(sample_indices)
    assert exact.sum() > n // 4
    assert np.abs(a[exact] - b[exact]).max() <= 1e-6
    assert ctx.rejected_instance_count() == 0
    for handle in handles:
        ctx.unregister_clip(handle)
    ctx.unregister_database(database)


def test_batch_launches_can_be_captured_into_a_hip_graph(setup):
    """A frame of decodes (transform poses + scalar curves) captured once into a hipGraph and replayed: the batch entry points only
    enqueue kernels on the caller's stream, so stream capture sees all of the work."""
    ctx, torch, device = setup
    clip = synth.build_clip(seed=2, num_tracks=100, num_samples=301, sample_rate=30.0)
    curves = synth.build_scalar_clip(seed=9, track_type=0, num_tracks=64, num_samples=90)
    handle, curve_handle = ctx.register_clip(clip.blob), ctx.register_clip(curves.blob)
    n = 4096
    rng = np.random.default_rng(41)
    d_clips = torch.full((n,), handle, dtype=torch.int32, device=device)
    d_curve_clips = torch.full((n,), curve_handle, dtype=torch.int32, device=device)
    d_times = torch.zeros(n, dtype=torch.float32, device=device)
    d_poses = torch.zeros((n, 100, 12), dtype=torch.float32, device=device)
    d_values = torch.zeros((n, 64), dtype=torch.float32, device=device)
    # the same frame also takes object space poses (pose consumers) and poses stored in other rows
    parents = synth.humanoid_hierarchy(100)
    ctx.set_clip_hierarchy(handle, parents)
    consumers = runtime.PoseConsumers()
    consumers.object_space = 1
    d_object_poses = torch.zeros((n, 100, 12), dtype=torch.float32, device=device)
    rows = rng.permutation(n).astype(np.int32)
    d_rows = torch.from_numpy(rows).to(device)
    d_scattered = torch.zeros((n, 100, 12), dtype=torch.float32, device=device)

    # ... and a decode in the order aclhip_order_instances_device gives, computed inside the graph (its per stream counters are allocated
    # by the first call on a stream: one call before the capture)
    d_order = torch.zeros(n, dtype=torch.int32, device=device)
    d_ordered_clips = torch.zeros(n, dtype=torch.int32, device=device)
    d_ordered_times = torch.zeros(n, dtype=torch.float32, device=device)
    d_ordered_poses = torch.zeros((n, 100, 12), dtype=torch.float32, device=device)

    graph = torch.cuda.CUDAGraph()
    capture_stream = torch.cuda.Stream(device)
    torch.cuda.synchronize(device)
    ctx.order_instances_device(d_clips.data_ptr(), d_times.data_ptr(), n, d_order.data_ptr(), d_ordered_clips.data_ptr(), d_ordered_times.data_ptr(), stream=capture_stream.cuda_stream)
    capture_stream.synchronize()
    with torch.cuda.graph(graph, stream=capture_stream):
        stream_handle = torch.cuda.current_stream(device).cuda_stream
        assert stream_handle == capture_stream.cuda_stream
        ctx.order_instances_device(d_clips.data_ptr(), d_times.data_ptr(), n, d_order.data_ptr(), d_ordered_clips.data_ptr(), d_ordered_times.data_ptr(), stream=stream_handle)
        ctx.decompress_tracks_batch(d_ordered_clips.data_ptr(), d_ordered_times.data_ptr(), n, d_ordered_poses.data_ptr(), 4800, stream=stream_handle)
        ctx.decompress_tracks_batch(d_clips.data_ptr(), d_times.data_ptr(), n, d_poses.data_ptr(), 4800, stream=stream_handle)
        ctx.decompress_scalar_tracks_batch(d_curve_clips.data_ptr(), d_times.data_ptr(), n, d_values.data_ptr(), 256, stream=stream_handle)
        ctx.decompress_poses_batch(d_clips.data_ptr(), d_times.data_ptr(), n, d_object_poses.data_ptr(), 4800, consumers, stream=stream_handle)
        ctx.decompress_tracks_batch_rows(d_clips.data_ptr(), d_times.data_ptr(), d_rows.data_ptr(), n, d_scattered.data_ptr(), 4800, stream=stream_handle)

    for frame in range(3):
        times = rng.uniform(0.0, min(clip.duration, curves.duration), size=n).astype(np.float32)
        d_times.copy_(torch.from_numpy(times))          # same buffers, new sample times: what an engine does every frame
        graph.replay()
        torch.cuda.synchronize(device)
        poses, values = d_poses.cpu().numpy(), d_values.cpu().numpy()
        assert helpers.exact(poses, ob.oracle_decompress_tracks_batch([clip.blob], np.zeros(n, dtype=np.uint32), times, 100))
        assert helpers.exact(values, ob.oracle_scalar_decompress_tracks_batch([curves.blob], np.zeros(n, dtype=np.uint32), times, 64))
        object_poses, scattered = d_object_poses.cpu().numpy(), d_scattered.cpu().numpy()
        assert helpers.exact(scattered[rows], poses)
        order = d_order.cpu().numpy()
        assert np.array_equal(np.sort(order), np.arange(n))
        assert helpers.exact(d_ordered_poses.cpu().numpy(), poses[order])
        assert helpers.bit_equal(object_poses, ob.oracle_decompress_poses_batch([clip.blob], np.zeros(n, dtype=np.uint32), times, 100, parent_indices=parents))
    assert ctx.rejected_instance_count() == 0
    ctx.unregister_clip(handle)
    ctx.unregister_clip(curve_handle)
