"""The real-compressor corpus (tests/golden/corpus: 208 compressed_tracks + 4 databases of 8 clips written by the reference's own
compress_track_list / build_database, tests/golden/make_corpus.py) through the C ABI on the GPU: every blob registers, and EVERY
SAMPLE x EVERY BONE of every clip decodes to the oracle's bits -- which tests/test_corpus_oracle.py holds to the reference's own decoder
on the same corpus. Mirrors the reference's only absolute check of this path, validate_accuracy
(/root/reference/tools/acl_compressor/sources/validate_tracks.cpp:92-260): debug settings at min(i / rate, duration) with the
nearest policy, decompress_track against decompress_tracks for every bone, the clamp / rounding / default-mode relations of :170-258;
and validate_db's progressive streaming, two chunks at a time (validate_database.cpp:44-103,499-679). Needs a GPU."""
import numpy as np
import pytest

from acl_amd import runtime
from oracle import bindings as ob
from oracle.database import OracleDatabase
import helpers

pytestmark = pytest.mark.gpu

CORPUS = helpers.load_corpus()


@pytest.fixture(scope="module")
def registered():
    """one context holding the whole corpus: (context, handles)"""
    context = runtime.Context(0)
    handles = [context.register_clip(clip["blob"]) for clip in CORPUS]
    yield context, handles
    assert context.rejected_instance_count() == 0
    for handle in handles:
        context.unregister_clip(handle)
    context.close()


def _batches(max_bytes=256 << 20):
    """the corpus' clips grouped so that one host pose buffer [instances, widest clip, 12] stays below max_bytes: (clip indices, widest)"""
    order = sorted(range(len(CORPUS)), key=lambda index: CORPUS[index]["spec"]["bones"])
    batches, current, instances = [], [], 0
    for index in order:
        spec = CORPUS[index]["spec"]
        count = 2 * spec["samples"] + 8
        if current and (instances + count) * spec["bones"] * 48 > max_bytes:
            batches.append(current)
            current, instances = [], 0
        current.append(index)
        instances += count
    if current:
        batches.append(current)
    return batches


def _times_of(clip):
    """every sample of the clip at min(i / rate, duration) (validate_tracks.cpp:217-219), and the points 0.37 of a sample further on
    (both keys of every pair, really interpolated)"""
    times, duration = helpers.corpus_sample_times(clip["blob"])
    rate = np.float32(clip["spec"]["rate"])
    between = np.minimum(times + np.float32(0.37) / rate, np.float32(duration)).astype(np.float32)
    return np.concatenate([times, between]), duration


def test_every_blob_registers_and_reports_its_shape(registered):
    context, handles = registered
    assert len(handles) == len(CORPUS) >= 200
    for clip, handle in zip(CORPUS, handles):
        info = context.clip_info(handle)
        assert info.num_tracks == clip["spec"]["bones"]
        assert info.duration == ob.oracle().aclo_finite_duration(clip["blob"].ctypes.data, ob.LOOP_AS_COMPRESSED)
        assert context.clip_matches(handle, clip["blob"])


@pytest.mark.parametrize("settings,policy", [(0, ob.ROUND_NEAREST), (0, ob.ROUND_NONE), (1, ob.ROUND_NEAREST), (1, ob.ROUND_PER_TRACK), (5, ob.ROUND_NONE)])
def test_every_sample_of_every_bone_equals_the_oracle(registered, settings, policy):
    """settings: 0 = the library's defaults (normalization lerp_only), 1 = debug_transform_decompression_settings (always + per track
    rounding: what validate_accuracy decodes with), 5 = never normalize"""
    context, handles = registered
    checked = 0
    for batch in _batches():
        widest = max(CORPUS[index]["spec"]["bones"] for index in batch)
        clips, times, blobs, which = [], [], [], []
        for slot, index in enumerate(batch):
            clip_times, _ = _times_of(CORPUS[index])
            clips.append(np.full(clip_times.size, handles[index], dtype=np.uint32))
            which.append(np.full(clip_times.size, slot, dtype=np.uint32))
            times.append(clip_times)
            blobs.append(CORPUS[index]["blob"])
        clips, times, which = np.concatenate(clips), np.concatenate(times), np.concatenate(which)
        if times.size == 0:
            continue
        rng = np.random.default_rng(77)
        track_rounding = rng.integers(0, 4, size=widest).astype(np.uint8) if settings == 1 else None
        params = helpers.gpu_params(runtime, rounding=policy, settings=settings)
        got = context.decompress_tracks(clips, times, params=params, num_tracks=widest, track_rounding=track_rounding)
        expected = ob.oracle_decompress_tracks_batch(blobs, which, times, widest, rounding=policy, options=helpers.oracle_options(settings, 0, None, track_rounding))
        for slot, index in enumerate(batch):
            rows = which == slot
            bones = CORPUS[index]["spec"]["bones"]
            assert helpers.bit_equal(got[rows][:, :bones], expected[rows][:, :bones]), \
                f"{CORPUS[index]['name']}: settings {settings} policy {policy}: {helpers.max_abs_diff(got[rows][:, :bones], expected[rows][:, :bones])}"
            checked += int(rows.sum()) * bones
    assert checked > 900_000           # transforms compared, bit for bit (927 906 with the committed corpus)


def test_decompress_track_of_every_sample_and_bone(registered):
    """validate_tracks.cpp:231-258: decompress_track against decompress_tracks for every bone of every sample -- translations and scales
    exactly, rotations within 1e-4 there; here all three bit for bit -- and against the oracle's decompress_track"""
    context, handles = registered
    requests = 0
    for batch in _batches(max_bytes=64 << 20):
        widest = max(CORPUS[index]["spec"]["bones"] for index in batch)
        clips, times, which = [], [], []
        for slot, index in enumerate(batch):
            clip_times, _ = helpers.corpus_sample_times(CORPUS[index]["blob"])
            clips.append(np.full(clip_times.size, handles[index], dtype=np.uint32))
            which.append(np.full(clip_times.size, slot, dtype=np.uint32))
            times.append(clip_times)
        clips, times, which = np.concatenate(clips), np.concatenate(times), np.concatenate(which)
        if times.size == 0:
            continue
        params = runtime.default_params(rounding_policy=ob.ROUND_NEAREST)
        poses = context.decompress_tracks(clips, times, params=params, num_tracks=widest)
        bones_of = np.array([CORPUS[index]["spec"]["bones"] for index in batch], dtype=np.int64)[which]
        instance = np.repeat(np.arange(times.size), bones_of)
        track = np.concatenate([np.arange(count) for count in bones_of]).astype(np.uint32)
        single = context.decompress_track(clips[instance], times[instance], track, params=runtime.default_params(rounding_policy=ob.ROUND_NEAREST))
        whole = poses[instance, track]
        assert helpers.bit_equal(single, whole), helpers.max_abs_diff(single, whole)
        requests += int(track.size)
        # the oracle's own single track decode (decompress_track_v0 restated) on every 97th request
        options = helpers.oracle_options(0)
        for k in range(0, track.size, 97):
            expected = ob.oracle_decompress_track(CORPUS[batch[int(which[instance[k]])]]["blob"], float(times[instance[k]]), int(track[k]), ob.ROUND_NEAREST, options)
            assert helpers.bit_equal(single[k], expected)
    assert requests > 450_000          # 463 953 with the committed corpus: every (sample, bone) of every clip


def test_clamping_rounding_and_default_mode_relations(registered):
    """validate_tracks.cpp:170-214,221-229 on the GPU's own outputs, all clips in one batch per relation:
       * a seek before the start / past the end decodes what a seek to the start / end does;
       * the per track rounding writer asking for policy P on every track == seeking with P (not for stripped clips, :197-200);
       * constant and variable default sub-track modes fed the clip's own defaults == the skipped mode over a buffer pre-filled with them"""
    context, handles = registered
    small = [index for index in range(len(CORPUS)) if CORPUS[index]["spec"]["bones"] <= 104]
    widest = max(CORPUS[index]["spec"]["bones"] for index in small)
    nearest = ob.ROUND_NEAREST

    def decode(clips, times, **options):
        track_rounding = options.pop("track_rounding", None)
        default_values = options.pop("default_values", None)
        out = options.pop("out", None)
        return context.decompress_tracks(np.asarray(clips, dtype=np.uint32), np.asarray(times, dtype=np.float32), params=helpers.gpu_params(runtime, **options), num_tracks=widest,
                                         track_rounding=track_rounding, default_values=default_values, out=out)

    clips = np.array([handles[index] for index in small], dtype=np.uint32)
    durations = np.array([context.clip_info(handles[index]).duration for index in small], dtype=np.float32)
    rates = np.array([CORPUS[index]["spec"]["rate"] for index in small], dtype=np.float32)
    # clamping
    assert helpers.bit_equal(decode(clips, np.full(clips.size, -0.2), rounding=nearest, settings=1), decode(clips, np.zeros(clips.size), rounding=nearest, settings=1))
    assert helpers.bit_equal(decode(clips, durations + 1.0, rounding=nearest, settings=1), decode(clips, durations, rounding=nearest, settings=1))
    # all rounding modes per track, at five times
    stripped = np.array([context.clip_info(handles[index]).has_stripped_keyframes != 0 for index in small])
    for fraction in (0.0, 0.2, 0.5, 0.75, 1.0):
        for policy in (ob.ROUND_NONE, ob.ROUND_FLOOR, ob.ROUND_CEIL, ob.ROUND_NEAREST):
            keep = ~stripped if policy != ob.ROUND_NONE else np.ones(clips.size, dtype=bool)
            seeked = decode(clips[keep], durations[keep] * fraction, rounding=policy, settings=1)
            per_track = decode(clips[keep], durations[keep] * fraction, rounding=ob.ROUND_PER_TRACK, settings=1, track_rounding=np.full(widest, policy, dtype=np.uint8))
            assert helpers.bit_equal(seeked, per_track), (fraction, policy, helpers.max_abs_diff(seeked, per_track))
    # default sub-track modes: skipped over a pre-filled buffer == constant / variable fed the same values
    times = durations * 0.37 + 0.25 / rates
    times = np.minimum(times, durations)
    constant = np.zeros((1, 12), dtype=np.float32)
    constant[0, 3] = 1.0
    constant[0, 8:11] = 1.0
    prefilled = np.tile(constant.reshape(1, 1, 12), (clips.size, widest, 1)).astype(np.float32)
    skipped = decode(clips, times, rounding=nearest, settings=1, default_mode=1, out=prefilled.copy())
    as_constant = decode(clips, times, rounding=nearest, settings=1, default_mode=2, default_values=constant)
    as_variable = decode(clips, times, rounding=nearest, settings=1, default_mode=3, default_values=np.tile(constant, (widest, 1)))
    for row, index in enumerate(small):
        bones = CORPUS[index]["spec"]["bones"]
        assert helpers.bit_equal(skipped[row, :bones], as_constant[row, :bones]) and helpers.bit_equal(skipped[row, :bones], as_variable[row, :bones]), CORPUS[index]["name"]


@pytest.mark.parametrize("name", helpers.CORPUS_DATABASES)
def test_databases_stream_in_and_out_two_chunks_at_a_time(name):
    """validate_db (validate_database.cpp:44-103,499-679): medium tier in (2 chunks, then the rest), low tier in, medium out (2, then the
    rest), low out, low in first, ... -- every clip decoded at every sample after every request, against the oracle's database_context"""
    case = helpers.load_corpus_database(name)
    with runtime.Context(0) as context:
        medium = case["bulk_medium"] if case["bulk_medium"].size else None
        low = case["bulk_low"] if case["bulk_low"].size else None
        database = context.register_database(case["database"], medium, low)
        clips = [context.register_clip_with_database(blob, database) for blob in case["clips"]]
        oracle_db = OracleDatabase(case["database"], case["bulk_medium"], case["bulk_low"])
        widest = max(ob.oracle().aclo_num_tracks(blob.ctypes.data) for blob in case["clips"])
        handles, times, which = [], [], []
        for slot, blob in enumerate(case["clips"]):
            clip_times, _ = helpers.corpus_sample_times(blob)
            handles.append(np.full(clip_times.size, clips[slot], dtype=np.uint32)), times.append(clip_times), which.append(np.full(clip_times.size, slot))
        handles, times, which = np.concatenate(handles), np.concatenate(times), np.concatenate(which)

        def check(state):
            for policy in (ob.ROUND_NEAREST, ob.ROUND_NONE):
                got = context.decompress_tracks(handles, times, params=runtime.default_params(rounding_policy=policy), num_tracks=widest)
                for slot, blob in enumerate(case["clips"]):
                    bones = ob.oracle().aclo_num_tracks(blob.ctypes.data)
                    rows = np.nonzero(which == slot)[0]
                    for row in rows[:: max(1, rows.size // 48)]:
                        expected = oracle_db.decompress_tracks(blob, float(times[row]), policy)
                        assert helpers.bit_equal(got[row, :bones], expected), f"{name}: {state}: clip {slot} t {times[row]}"

        def stream(tier, stream_in, num_chunks):
            moved = (context.database_stream_in if stream_in else context.database_stream_out)(database, tier, num_chunks)
            expected = (oracle_db.stream_in if stream_in else oracle_db.stream_out)(tier, num_chunks)
            assert moved == expected
            assert list(context.database_info(database).num_loaded_chunks) == [sum(oracle_db.loaded[1]), sum(oracle_db.loaded[2])]

        everything = 0xFFFFFFFF
        check("nothing streamed in")
        for first, second in ((runtime.TIER_MEDIUM_IMPORTANCE, runtime.TIER_LOWEST_IMPORTANCE), (runtime.TIER_LOWEST_IMPORTANCE, runtime.TIER_MEDIUM_IMPORTANCE)):
            for tier in (first, second):
                stream(tier, True, 2)
                check(f"tier {tier}: two chunks in")
                stream(tier, True, everything)
                check(f"tier {tier}: all in")
            for tier in (first, second):
                stream(tier, False, 2)
                check(f"tier {tier}: two chunks out")
                stream(tier, False, everything)
                check(f"tier {tier}: all out")
        assert context.rejected_instance_count() == 0
        for clip in clips:
            context.unregister_clip(clip)
        context.unregister_database(database)


def test_hierarchy_comes_from_the_blobs_own_metadata(registered):
    """aclhip_set_clip_hierarchy_from_metadata: local -> object space of a clip compressed with include_parent_track_indices /
    include_track_descriptions without the caller passing its skeleton == the same clip given the skeleton by hand == the oracle's
    local_to_object_space over the oracle's local pose (held to oracle/_ref/libaclref_pose.so by tests/test_pose_consumers_oracle.py)"""
    context, handles = registered
    carried = [index for index, clip in enumerate(CORPUS) if clip["spec"].get("include_parent_track_indices") or clip["spec"].get("include_track_descriptions")]
    assert len(carried) >= 3
    for index in carried:
        clip, handle = CORPUS[index], handles[index]
        info = context.clip_metadata_info(handle)
        assert info.has_metadata == 1 and info.has_parent_track_indices == 1
        parents = context.clip_parent_indices(handle)
        assert np.array_equal(parents, clip["parents"].astype(np.uint32))
        times, _ = _times_of(clip)
        context.set_clip_hierarchy_from_metadata(handle)
        from_metadata = context.decompress_poses(np.full(times.size, handle, dtype=np.uint32), times, object_space=True)
        by_hand_handle = context.register_clip(clip["blob"])
        context.set_clip_hierarchy(by_hand_handle, clip["parents"].astype(np.uint32))
        by_hand = context.decompress_poses(np.full(times.size, by_hand_handle, dtype=np.uint32), times, object_space=True)
        context.unregister_clip(by_hand_handle)
        assert helpers.bit_equal(from_metadata, by_hand)
        for row in range(0, times.size, max(1, times.size // 12)):
            local = ob.oracle_decompress_tracks(clip["blob"], float(times[row]))
            assert helpers.bit_equal(from_metadata[row], ob.oracle_local_to_object_space(clip["parents"].astype(np.uint32), local)), clip["name"]
    # a clip that carries no parent indices says so
    bare = next(index for index, clip in enumerate(CORPUS) if index not in carried)
    assert context.clip_metadata_info(handles[bare]).has_parent_track_indices == 0
    with pytest.raises(runtime.AclHipError) as error:
        context.set_clip_hierarchy_from_metadata(handles[bare])
    assert error.value.status == runtime.ERROR_NO_METADATA
    with pytest.raises(runtime.AclHipError) as error:
        context.clip_track_descriptions(handles[bare])
    assert error.value.status == runtime.ERROR_NO_METADATA


def test_bind_pose_default_mode_takes_the_blobs_own_track_descriptions(registered):
    """ACLHIP_DEFAULT_BIND_POSE: default sub-tracks decode to track_desc_transformf::default_value of their track, read from the blob --
    == the variable mode fed that table by hand (GPU and oracle); clips without descriptions fall back on the identity"""
    context, handles = registered
    bind = runtime.DEFAULT_BIND_POSE
    described = [index for index, clip in enumerate(CORPUS) if clip["spec"].get("include_track_descriptions")]
    assert len(described) >= 3 and any(CORPUS[index]["bind_is_default"] for index in described)
    for index in described + [next(i for i, clip in enumerate(CORPUS) if clip["bind_is_default"] and i not in described)]:
        clip, handle = CORPUS[index], handles[index]
        bones = clip["spec"]["bones"]
        if index in described:
            table, _, _ = context.clip_track_descriptions(handle)
        else:
            table = np.tile(np.array([0, 0, 0, 1, 0, 0, 0, 0, 1, 1, 1, 0], dtype=np.float32), (bones, 1))     # no descriptions: qvv_identity
        times, _ = _times_of(clip)
        clips = np.full(times.size, handle, dtype=np.uint32)
        for settings in (0, 1):
            normalization, per_track = helpers.settings_pair(settings)
            modes = dict(default_rotation_mode=bind, default_translation_mode=bind, default_scale_mode=bind, normalization=normalization, per_track_rounding=per_track)
            got = context.decompress_tracks(clips, times, params=runtime.default_params(**modes))
            by_hand = context.decompress_tracks(clips, times, params=helpers.gpu_params(runtime, settings=settings, default_mode=3), default_values=table)
            assert helpers.bit_equal(got, by_hand), clip["name"]
            expected = ob.oracle_decompress_tracks_batch([clip["blob"]], np.zeros(times.size, dtype=np.uint32), times, bones, options=helpers.oracle_options(settings, 3, table))
            assert helpers.bit_equal(got, expected), clip["name"]
            # mixed modes: rotations from the bind pose, translations skipped (pre-filled), scales constant
            mixed = dict(modes, default_translation_mode=runtime.DEFAULT_SKIPPED, default_scale_mode=runtime.DEFAULT_CONSTANT)
            prefilled = np.full((times.size, bones, 12), 7.0, dtype=np.float32)
            got_mixed = context.decompress_tracks(clips, times, params=runtime.default_params(**mixed), out=prefilled.copy())
            options = helpers.oracle_options(settings, 3, table)
            options.default_translation_mode, options.default_scale_mode = ob.DEFAULT_SKIPPED, ob.DEFAULT_CONSTANT
            options.default_values = table.ctypes.data          # (constant mode reads row 0 of the table: the oracle's and the library's same rule)
            expected_mixed = prefilled.copy()
            for row in range(times.size):
                ob.oracle_decompress_tracks(clip["blob"], float(times[row]), ob.ROUND_NONE, options, out=expected_mixed[row])
            if index not in described or np.array_equal(table[0, 8:11], np.ones(3, dtype=np.float32)):
                assert helpers.bit_equal(got_mixed, expected_mixed), clip["name"]
        # single track requests follow the same table
        rng = np.random.default_rng(5)
        tracks = rng.integers(0, bones, size=times.size).astype(np.uint32)
        single = context.decompress_track(clips, times, tracks, params=runtime.default_params(default_rotation_mode=bind, default_translation_mode=bind, default_scale_mode=bind))
        whole = context.decompress_tracks(clips, times, params=runtime.default_params(default_rotation_mode=bind, default_translation_mode=bind, default_scale_mode=bind))
        assert helpers.bit_equal(single, whole[np.arange(times.size), tracks])


def test_bind_pose_defaults_in_single_track_requests_of_mixed_waves(registered):
    """waves of requests that name DIFFERENT clips gather their clip records four lanes per record and fetch the records' second halves
    -- where the bind pose hangs -- only for launches with table defaults (kernels_track.inl: gather_clip_records): every request
    == its row of the clip's whole pose under ACLHIP_DEFAULT_BIND_POSE, and under the variable mode with one caller table"""
    context, handles = registered
    bind = runtime.DEFAULT_BIND_POSE
    described = [index for index, clip in enumerate(CORPUS) if clip["spec"].get("include_track_descriptions")]
    others = [index for index, clip in enumerate(CORPUS) if index not in described and clip["spec"]["bones"] >= 4][:6]
    chosen = described[:6] + others
    rng = np.random.default_rng(17)
    which = rng.integers(0, len(chosen), size=700)
    bones = np.array([CORPUS[index]["spec"]["bones"] for index in chosen])
    durations = np.array([ob.oracle().aclo_finite_duration(CORPUS[index]["blob"].ctypes.data, ob.LOOP_AS_COMPRESSED) for index in chosen], dtype=np.float32)
    times = (rng.uniform(0.0, 1.0, size=which.size) * durations[which]).astype(np.float32)
    tracks = (rng.uniform(0.0, 1.0, size=which.size) * bones[which]).astype(np.uint32)
    clips = np.array([handles[index] for index in chosen], dtype=np.uint32)[which]
    table = rng.uniform(-1.0, 1.0, size=(int(bones.max()), 12)).astype(np.float32)
    for params, defaults in ((runtime.default_params(default_rotation_mode=bind, default_translation_mode=bind, default_scale_mode=bind), None),
                             (helpers.gpu_params(runtime, settings=0, default_mode=3), table)):
        single = context.decompress_track(clips, times, tracks, params=params, **({"default_values": defaults} if defaults is not None else {}))
        for k, index in enumerate(chosen):
            rows = np.flatnonzero(which == k)
            whole = context.decompress_tracks(np.full(rows.size, handles[index], dtype=np.uint32), times[rows], params=params, **({"default_values": defaults} if defaults is not None else {}))
            assert helpers.bit_equal(single[rows], whole[np.arange(rows.size), tracks[rows]]), CORPUS[index]["name"]
    assert context.rejected_instance_count() == 0
