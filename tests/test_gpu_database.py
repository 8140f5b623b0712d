"""Streamed keyframe tiers (compressed_database) through the C ABI: parity with the reference's golden vectors and with the CPU
oracle after every stream_in / stream_out request. Needs a GPU."""
import os

import numpy as np
import pytest

from acl_amd import runtime, synth
from oracle import bindings as ob
from oracle.database import OracleDatabase
import helpers

pytestmark = pytest.mark.gpu


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


def _register(context, case, inline):
    if inline:
        database = context.register_database(case["database_inline"])
    else:
        database = context.register_database(case["database"], case["bulk_medium"], case["bulk_low"])
    clips = [context.register_clip_with_database(clip, database) for clip in case["clips"]]
    return database, clips


def _release(context, database, clips):
    for clip in clips:
        context.unregister_clip(clip)
    context.unregister_database(database)


@pytest.mark.parametrize("name", helpers.database_golden_cases())
@pytest.mark.parametrize("inline", [False, True])
def test_streaming_states_match_reference_golden(context, name, inline):
    case = helpers.load_database_golden(name)
    database, clips = _register(context, case, inline)
    info = context.database_info(database)
    assert info.num_clips == len(clips)
    num_times = case["times"].shape[1]
    max_tracks = case["poses"].shape[4]

    def check(state):
        # one batch holds every clip at every time: instances of different clips share the launch
        handles = np.repeat(np.array(clips, dtype=np.uint32), num_times)
        times = case["times"].reshape(-1)
        for p, policy in enumerate(case["policies"]):
            poses = context.decompress_tracks(handles, times, params=runtime.default_params(rounding_policy=int(policy)), num_tracks=max_tracks)
            for c, clip in enumerate(case["clips"]):
                num_tracks = ob.oracle().aclo_num_tracks(clip.ctypes.data)
                actual = poses[c * num_times: (c + 1) * num_times, :num_tracks]
                expected = case["poses"][state, c, p, :, :num_tracks]
                assert helpers.bit_equal(actual, expected), f"{name}: state {state} clip {c} policy {policy}: {helpers.max_abs_diff(actual, expected)}"

    check(0)
    for state, (tier, num_chunks, stream_in) in enumerate(case["ops"]):
        stream = context.database_stream_in if stream_in else context.database_stream_out
        moved = stream(database, int(tier), int(num_chunks))
        assert (moved != 0) == (case["results"][state] == 1)
        check(state + 1)
    assert context.rejected_instance_count() == 0
    _release(context, database, clips)


def test_single_track_requests_follow_the_tiers(context):
    case = helpers.load_database_golden("three_clips_4k_chunks")
    database, clips = _register(context, case, False)
    oracle_db = OracleDatabase(case["database"], case["bulk_medium"], case["bulk_low"])
    rng = np.random.default_rng(3)
    for tier, num_chunks in ((None, 0), (1, 2), (2, 1), (2, 0xFFFFFFFF), (1, 0xFFFFFFFF)):
        if tier is not None:
            assert context.database_stream_in(database, tier, num_chunks) == oracle_db.stream_in(tier, num_chunks)
        assert list(context.database_info(database).num_loaded_chunks) == [sum(oracle_db.loaded[1]), sum(oracle_db.loaded[2])]
        for c, clip in enumerate(case["clips"]):
            num_tracks = ob.oracle().aclo_num_tracks(clip.ctypes.data)
            duration = ob.oracle().aclo_finite_duration(clip.ctypes.data, ob.LOOP_AS_COMPRESSED)
            times = rng.uniform(0.0, duration, size=64).astype(np.float32)
            tracks = rng.integers(0, num_tracks, size=64).astype(np.uint32)
            out = context.decompress_track(np.full(64, clips[c], dtype=np.uint32), times, tracks)
            whole = context.decompress_tracks(np.full(64, clips[c], dtype=np.uint32), times)
            for i in range(64):
                expected = oracle_db.decompress_tracks(clip, float(times[i]))
                assert helpers.bit_equal(whole[i], expected)
                assert helpers.bit_equal(out[i], expected[tracks[i]])
    _release(context, database, clips)


def test_single_track_requests_of_mixed_waves_follow_the_tiers(context):
    """waves whose requests name DIFFERENT clips -- database bound ones next to plain ones -- gather their clip records four lanes per
    record and fetch the records' second halves (the tiers) only when some request of the wave is bound to a database
    (kernels_track.inl: gather_clip_records): every request against the restated database_context / the oracle, in every residency"""
    case = helpers.load_database_golden("three_clips_4k_chunks")
    database, clips = _register(context, case, False)
    oracle_db = OracleDatabase(case["database"], case["bulk_medium"], case["bulk_low"])
    plain = [synth.build_clip(seed=40 + i, num_tracks=9 + 7 * i, num_samples=25 + 20 * i, strip_keyframes=i % 2) for i in range(3)]
    plain_handles = [context.register_clip(clip.blob) for clip in plain]
    blobs = list(case["clips"]) + [clip.blob for clip in plain]
    handles = np.array(list(clips) + plain_handles, dtype=np.uint32)
    num_tracks = np.array([ob.oracle().aclo_num_tracks(blob.ctypes.data) for blob in blobs])
    durations = np.array([ob.oracle().aclo_finite_duration(blob.ctypes.data, ob.LOOP_AS_COMPRESSED) for blob in blobs], dtype=np.float32)
    rng = np.random.default_rng(11)
    for tier, num_chunks in ((None, 0), (1, 1), (2, 2), (1, 0xFFFFFFFF), (2, 0xFFFFFFFF)):
        if tier is not None:
            assert context.database_stream_in(database, tier, num_chunks) == oracle_db.stream_in(tier, num_chunks)
        for which in (rng.integers(0, handles.size, size=333), rng.integers(3, handles.size, size=130), np.repeat(rng.integers(0, handles.size, size=5), 64)):
            times = (rng.uniform(0.0, 1.0, size=which.size) * durations[which]).astype(np.float32)
            tracks = (rng.uniform(0.0, 1.0, size=which.size) * num_tracks[which]).astype(np.uint32)
            out = context.decompress_track(handles[which], times, tracks)
            for i in range(which.size):
                blob = blobs[which[i]]
                expected = oracle_db.decompress_tracks(blob, float(times[i])) if which[i] < len(clips) else ob.oracle_decompress_tracks_batch([blob], np.zeros(1, dtype=np.uint32), times[i: i + 1], int(num_tracks[which[i]]))[0]
                assert helpers.bit_equal(out[i], expected[tracks[i]]), (tier, i, which[i])
    assert context.rejected_instance_count() == 0
    for handle in plain_handles:
        context.unregister_clip(handle)
    _release(context, database, clips)


def test_database_bound_clip_without_database_uses_only_its_own_keyframes(context):
    """decompression_context::initialize(tracks) of a clip that was split into a database: legal, lowest quality
    (impl/decompress.impl.h:58-83 leaves db = nullptr; seek falls back to the clip's sample_indices)."""
    case = helpers.load_database_golden("two_clips_single_chunk")
    for c, clip in enumerate(case["clips"]):
        handle = context.register_clip(clip)
        assert context.clip_info(handle).has_database == 1
        poses = context.decompress_tracks(np.full(case["times"].shape[1], handle, dtype=np.uint32), case["times"][c])
        num_tracks = ob.oracle().aclo_num_tracks(clip.ctypes.data)
        assert helpers.bit_equal(poses[:, :num_tracks], case["poses"][0, c, 0, :, :num_tracks])
        context.unregister_clip(handle)


def test_database_error_paths(context):
    case = helpers.load_database_golden("two_clips_single_chunk")
    other = helpers.load_database_golden("medium_tier_only")
    database = context.register_database(case["database"], case["bulk_medium"], case["bulk_low"])

    with pytest.raises(runtime.AclHipError) as error:      # compressed_database::contains() is false
        context.register_clip_with_database(other["clips"][0], database)
    assert error.value.status == 9
    with pytest.raises(runtime.AclHipError) as error:      # a clip that was never split into a database
        context.register_clip_with_database(synth.build_clip(seed=5, num_tracks=8, num_samples=20).blob, database)
    assert error.value.status == 9
    with pytest.raises(runtime.AclHipError) as error:
        context.register_clip_with_database(case["clips"][0], 12345)
    assert error.value.status == 8
    with pytest.raises(runtime.AclHipError):               # quality_tier::highest_importance lives in the clip
        context.database_stream_in(database, 0, 1)
    with pytest.raises(runtime.AclHipError):               # bulk data neither inline nor given
        context.register_database(case["database"])
    corrupt = case["bulk_medium"].copy()
    corrupt[40] ^= 0xFF
    with pytest.raises(runtime.AclHipError):               # bulk data hash
        context.register_database(case["database"], corrupt, case["bulk_low"])
    truncated = case["database"][:40].copy()
    with pytest.raises(runtime.AclHipError):
        context.register_database(truncated, case["bulk_medium"], case["bulk_low"])

    clip = context.register_clip_with_database(case["clips"][0], database)
    with pytest.raises(runtime.AclHipError):               # clips still bound
        context.unregister_database(database)
    context.unregister_clip(clip)
    context.unregister_database(database)
    with pytest.raises(runtime.AclHipError):
        context.database_info(database)


def test_stream_in_is_ordered_with_decodes_on_the_same_stream(context):
    """A decode enqueued behind a stream_in on one stream sees the new tier; no host synchronisation in between."""
    import torch
    case = helpers.load_database_golden("three_clips_4k_chunks")
    database, clips = _register(context, case, False)
    oracle_db = OracleDatabase(case["database"], case["bulk_medium"], case["bulk_low"])
    clip_index = 2
    blob = case["clips"][clip_index]
    num_tracks = ob.oracle().aclo_num_tracks(blob.ctypes.data)
    duration = ob.oracle().aclo_finite_duration(blob.ctypes.data, ob.LOOP_AS_COMPRESSED)
    times = np.linspace(0.0, duration, 512).astype(np.float32)
    d_clips = torch.full((times.size,), clips[clip_index], dtype=torch.int32, device="cuda")
    d_times = torch.from_numpy(times).cuda()
    stream = torch.cuda.Stream()
    stream.wait_stream(torch.cuda.current_stream())         # d_clips / d_times were uploaded on the current stream
    snapshots = []
    with torch.cuda.stream(stream):
        for tier in (None, 1, 2):
            if tier is not None:
                context.database_stream_in(database, tier, stream=stream.cuda_stream)
            poses = torch.zeros((times.size, num_tracks, 12), dtype=torch.float32, device="cuda")
            context.decompress_tracks_batch(d_clips.data_ptr(), d_times.data_ptr(), times.size, poses.data_ptr(), num_tracks * 48, stream=stream.cuda_stream)
            snapshots.append(poses)
    stream.synchronize()
    for tier, poses in zip((None, 1, 2), snapshots):
        if tier is not None:
            oracle_db.stream_in(tier)
        actual = poses.cpu().numpy()
        for i in range(times.size):
            assert helpers.bit_equal(actual[i], oracle_db.decompress_tracks(blob, float(times[i])))
    _release(context, database, clips)


def test_decodes_on_one_stream_race_stream_ins_on_another():
    """The tier metadata a decode reads lives in the clips' own sample records (database_sample_record), rewritten behind every
    stream_in / stream_out. While one stream decodes without pause another streams both tiers in and out, chunk by chunk: every pose
    of every launch is finite with unit rotations and every instance of a racing launch decodes to what SOME residency of the two tiers
    gives (a tier's metadata word is written and read whole: old or new, like the reference's relaxed atomics); once both streams have
    been joined the decode is exactly the oracle's for the final state, instance by instance. A clip bound while requests are still
    queued gets the state behind them."""
    import torch
    case = helpers.load_database_golden("three_clips_4k_chunks")
    with runtime.Context(0) as context:
        database = context.register_database(case["database"], case["bulk_medium"], case["bulk_low"])
        clips = [context.register_clip_with_database(clip, database) for clip in case["clips"][:-1]]
        late_blob = case["clips"][-1]
        info = context.database_info(database)
        rng = np.random.default_rng(5)
        num_tracks = [ob.oracle().aclo_num_tracks(clip.ctypes.data) for clip in case["clips"]]
        max_tracks = max(num_tracks)
        n = 2048
        which = rng.integers(0, len(clips), size=n)
        durations = np.array([ob.oracle().aclo_finite_duration(clip.ctypes.data, ob.LOOP_AS_COMPRESSED) for clip in case["clips"]], dtype=np.float32)
        times = (rng.uniform(0.0, 1.0, size=n).astype(np.float32) * durations[which]).astype(np.float32)
        d_clips = torch.from_numpy(np.array(clips, dtype=np.int32)[which]).cuda()
        d_times = torch.from_numpy(times).cuda()

        # what each instance decodes to under the four whole-tier residencies
        candidates = []
        for medium in (False, True):
            for low in (False, True):
                oracle_db = OracleDatabase(case["database"], case["bulk_medium"], case["bulk_low"])
                if medium:
                    oracle_db.stream_in(1)
                if low:
                    oracle_db.stream_in(2)
                candidates.append(np.stack([np.pad(oracle_db.decompress_tracks(case["clips"][c], float(t)), ((0, max_tracks - num_tracks[c]), (0, 0))) for c, t in zip(which, times)]))

        decode_stream, tier_stream = torch.cuda.Stream(), torch.cuda.Stream()
        decode_stream.wait_stream(torch.cuda.current_stream())
        launches = 96
        d_poses = torch.zeros((launches, n, max_tracks, 12), dtype=torch.float32, device="cuda")
        for k in range(launches):
            context.decompress_tracks_batch(d_clips.data_ptr(), d_times.data_ptr(), n, d_poses[k].data_ptr(), max_tracks * 48, stream=decode_stream.cuda_stream)
            # whole tiers move, one request each, while the decodes run (two requests in flight per decode launch on average)
            tier = 1 + (k // 2) % 2
            (context.database_stream_in if (k // 4) % 2 == 0 else context.database_stream_out)(database, tier, stream=tier_stream.cuda_stream)
        # a clip bound now: its records get the tiers' state BEHIND the requests still queued
        late = context.register_clip_with_database(late_blob, database)
        tier_stream.synchronize()
        decode_stream.synchronize()
        poses = d_poses.cpu().numpy()
        assert np.isfinite(poses).all()
        lengths = np.linalg.norm(poses[..., 0:4], axis=-1)
        live = np.arange(max_tracks)[None, :] < np.array(num_tracks)[which][:, None]
        assert np.abs(lengths[:, live] - 1.0).max() < 1.0e-4
        views = [np.ascontiguousarray(c[..., helpers.XYZ_LANES]).view(np.uint32) for c in candidates]
        for k in range(launches):
            got = np.ascontiguousarray(poses[k][..., helpers.XYZ_LANES]).view(np.uint32)
            matches = np.zeros(n, dtype=bool)
            for view in views:
                matches |= (got == view).reshape(n, -1).all(axis=1)
            # (an instance whose two keys lie in different segments may see one segment before and one after a request: rare, and still
            # a legal reading of two relaxed atomics; everything else is one of the four residencies)
            assert matches.mean() > 0.9, (k, matches.mean())

        # joined: the final state, exactly. Residency as the host saw the requests go through:
        oracle_db = OracleDatabase(case["database"], case["bulk_medium"], case["bulk_low"])
        for k in range(launches):
            tier = 1 + (k // 2) % 2
            (oracle_db.stream_in if (k // 4) % 2 == 0 else oracle_db.stream_out)(tier)
        all_clips = clips + [late]
        for c, handle in enumerate(all_clips):
            clip_times = np.linspace(0.0, float(durations[c]), 64).astype(np.float32)
            out = context.decompress_tracks(np.full(64, handle, dtype=np.uint32), clip_times)
            for i in range(64):
                assert helpers.bit_equal(out[i], oracle_db.decompress_tracks(case["clips"][c], float(clip_times[i]))), (c, i)
        assert context.rejected_instance_count() == 0
        for handle in all_clips:
            context.unregister_clip(handle)
        context.unregister_database(database)


def test_a_streamed_request_that_fails_half_way_leaves_nothing_behind():
    """Caller-supplied streamers hand over bulk data that arrives from outside: untrusted. A request whose LATER chunk is corrupt
    must leave no chunk parsed-but-not-uploaded (the patches of the chunks in front of it would never reach the device, and the
    kernel that applies them would read table entries nobody wrote): all of a request's chunks are checked before any state changes,
    and the same request with good data then streams everything in."""
    name = "three_clips_4k_chunks"
    case = helpers.load_database_golden(name)
    with runtime.Context(0) as context:
        database = context.register_database_streamed(case["database"])
        clips = [context.register_clip_with_database(clip, database) for clip in case["clips"]]
        info = context.database_info(database)
        oracle_db = OracleDatabase(case["database"], case["bulk_medium"], case["bulk_low"])
        num_times = case["times"].shape[1]
        handles = np.repeat(np.array(clips, dtype=np.uint32), num_times)
        times = case["times"].reshape(-1)
        max_tracks = case["poses"].shape[4]

        def check():
            poses = context.decompress_tracks(handles, times, num_tracks=max_tracks)
            for c, clip in enumerate(case["clips"]):
                num_tracks = ob.oracle().aclo_num_tracks(clip.ctypes.data)
                for t in range(num_times):
                    expected = oracle_db.decompress_tracks(clip, float(case["times"][c, t]))
                    assert helpers.bit_equal(poses[c * num_times + t, :num_tracks], expected), (c, t)

        for tier, bulk in ((runtime.TIER_MEDIUM_IMPORTANCE, case["bulk_medium"]), (runtime.TIER_LOWEST_IMPORTANCE, case["bulk_low"])):
            num_chunks = int(info.num_chunks[tier - 1])
            if num_chunks < 2 or bulk.size == 0:
                continue
            # the LAST chunk's header lies about its index: the request for all chunks must fail as a whole
            corrupt = synth.aligned_bytes(bulk.size)
            corrupt[:] = bulk
            offsets = [int(offset) for offset in oracle_db.chunks[tier][:, 1]]
            corrupt[offsets[-1]: offsets[-1] + 4] = np.frombuffer(np.uint32(0xDEAD).tobytes(), dtype=np.uint8)
            with pytest.raises(runtime.AclHipError):
                context.database_stream_in_from(database, tier, corrupt)
            check()                                                    # nothing became resident
            good = synth.aligned_bytes(bulk.size)
            good[:] = bulk
            assert context.database_stream_in_from(database, tier, good) == num_chunks
            oracle_db.stream_in(tier, num_chunks)
            check()
        assert context.rejected_instance_count() == 0
        for clip in clips:
            context.unregister_clip(clip)
        context.unregister_database(database)


def test_a_streamed_chunk_that_repeats_another_chunks_segment_is_refused():
    """the arrival side of tests/test_host_validation.py::test_two_chunks_with_keyframes_of_the_same_segment_are_refused: the second
    chunk of the medium tier names a segment header the chunks in front of it (or it) already patch -> the request is refused whole,
    the good bytes then stream in and decode to the restated database_context"""
    case = helpers.load_database_golden("three_clips_4k_chunks")
    bulk = case["bulk_medium"]
    assert bulk[4284] == 8
    with runtime.Context(0) as context:
        database = context.register_database_streamed(case["database"])
        clips = [context.register_clip_with_database(clip, database) for clip in case["clips"]]
        corrupt = synth.aligned_bytes(bulk.size)
        corrupt[:] = bulk
        corrupt[4284] = 24
        with pytest.raises(runtime.AclHipError, match="same segment"):
            context.database_stream_in_from(database, runtime.TIER_MEDIUM_IMPORTANCE, corrupt)
        good = synth.aligned_bytes(bulk.size)
        good[:] = bulk
        num_chunks = int(context.database_info(database).num_chunks[0])
        assert context.database_stream_in_from(database, runtime.TIER_MEDIUM_IMPORTANCE, good) == num_chunks
        oracle_db = OracleDatabase(case["database"], case["bulk_medium"], case["bulk_low"])
        oracle_db.stream_in(runtime.TIER_MEDIUM_IMPORTANCE, num_chunks)
        for blob, clip in zip(case["clips"], clips):
            duration = float(context.clip_info(clip).duration)
            times = np.linspace(0.0, duration, 9).astype(np.float32)
            poses = context.decompress_tracks(np.full(times.size, clip, dtype=np.uint32), times)
            for row, t in enumerate(times):
                assert helpers.bit_equal(poses[row], oracle_db.decompress_tracks(blob, float(t))), (row, t)
        for clip in clips:
            context.unregister_clip(clip)
        context.unregister_database(database)
