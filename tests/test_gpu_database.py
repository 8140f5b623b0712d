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
        for i in range(0, times.size, 17):
            assert helpers.bit_equal(actual[i], oracle_db.decompress_tracks(blob, float(times[i])))
    _release(context, database, clips)


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
