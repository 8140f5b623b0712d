"""aclhip_order_instances_for_locality / _for_pose_windows (host only, no GPU): the decode order is a permutation under which every XCD
-- workgroup b of a launch runs on XCD b % 8 and holds 4 consecutive (instance, pose window) wavefronts -- serves one contiguous share
of the instance list bucketed by clip: every clip on one XCD (two where a boundary cuts it), next to its other instances."""
import numpy as np
import pytest

from acl_amd import runtime

WAVES_PER_WORKGROUP = 4


def xcd_of_slots(num_slots, windows_per_instance):
    """the XCD on which the first wavefront of every slot of a launch runs"""
    return (np.arange(num_slots, dtype=np.int64) * windows_per_instance // WAVES_PER_WORKGROUP) % 8


def check_order(clips, order, windows_per_instance, stable=True):
    num_instances = clips.size
    assert np.array_equal(np.sort(order), np.arange(num_instances))
    ordered = clips[order].astype(np.int64)
    xcd = xcd_of_slots(num_instances, windows_per_instance)
    if windows_per_instance % 8 == 0:
        xcd = np.arange(num_instances) % 8      # such poses span every XCD: plain bucketing by clip, dealt out slot by slot
    pairs = set()
    previous_last_clip, previous_last_instance = -1, -1
    for x in range(8):
        mine = ordered[xcd == x]
        # an XCD sees its clips in ascending order: every clip's instances are neighbours in the XCD's own sequence of slots
        assert np.all(np.diff(mine) >= 0)
        # ... and the XCDs' shares are consecutive ranges of ONE bucketed sequence
        if mine.size:
            assert mine[0] >= previous_last_clip
        # stable: instances of one clip keep their relative order, inside a share and across the boundary that cuts the clip
        instances = order[xcd == x].astype(np.int64)
        same_clip = np.diff(mine) == 0
        assert not stable or np.all(np.diff(instances)[same_clip] > 0)
        if mine.size:
            if stable and mine[0] == previous_last_clip:
                assert instances[0] > previous_last_instance
            previous_last_clip, previous_last_instance = mine[-1], instances[-1]
        pairs.update((int(c), x) for c in np.unique(mine))
    # a clip is split over two XCDs only where a boundary between shares falls inside it
    assert len(pairs) <= np.unique(clips).size + 7


@pytest.mark.parametrize("windows_per_instance", [1, 2, 3, 4, 5, 8, 16])
@pytest.mark.parametrize("num_instances,num_clips", [(0, 1), (1, 1), (3, 2), (4, 1), (31, 5), (1000, 7), (65536, 256), (10001, 1000), (4096, 8)])
def test_order_is_a_locality_preserving_permutation(num_instances, num_clips, windows_per_instance):
    rng = np.random.default_rng(num_instances + num_clips)
    clips = rng.integers(0, num_clips, size=num_instances).astype(np.uint32)
    order = runtime.order_instances_for_locality(clips, windows_per_instance)
    check_order(clips, order, windows_per_instance)


def test_every_xcd_serves_an_equal_share_of_an_evenly_spread_batch():
    rng = np.random.default_rng(7)
    clips = rng.integers(0, 256, size=65536).astype(np.uint32)
    order = runtime.order_instances_for_locality(clips)
    xcd = xcd_of_slots(clips.size, 1)
    ordered = clips[order]
    for x in range(8):
        mine = np.unique(ordered[xcd == x])
        assert 32 <= mine.size <= 34        # 256 clips over 8 XCDs, boundaries cut at most two of an XCD's clips


def test_skewed_batches_stay_balanced():
    """90 % of the instances on one clip: it is dealt to as many XCDs as it fills, nothing is dropped, the rest stays bucketed"""
    rng = np.random.default_rng(1)
    clips = np.where(rng.uniform(size=20000) < 0.9, 5, rng.integers(0, 64, size=20000)).astype(np.uint32)
    order = runtime.order_instances_for_locality(clips)
    assert np.array_equal(np.sort(order), np.arange(clips.size))
    xcd = xcd_of_slots(clips.size, 1)
    for x in range(8):
        assert np.all(np.diff(clips[order][xcd == x].astype(np.int64)) >= 0)


def test_arbitrary_handle_values_take_the_comparison_sort():
    rng = np.random.default_rng(3)
    clips = rng.choice(np.array([7, 4000000000, 123456789, 99, 2 ** 31], dtype=np.uint32), size=777)
    order = runtime.order_instances_for_locality(clips)
    check_order(clips, order, 1)


def test_null_arguments():
    lib = runtime.load_library()
    assert lib.aclhip_order_instances_for_locality(None, None, 4, None) == runtime.ERROR_INVALID_ARGUMENT
    assert lib.aclhip_order_instances_for_locality(None, None, 0, None) == runtime.OK
    assert lib.aclhip_order_instances_for_pose_windows(0, None, 0, None) == runtime.ERROR_INVALID_ARGUMENT


@pytest.mark.parametrize("num_requests", [1, 5, 255, 256, 257, 2047, 2048, 4099, 70001])
def test_track_request_order_is_a_stable_bucketing_by_clip_with_one_xcd_per_clip(num_requests):
    """aclhip_order_track_requests_for_locality: workgroup b of aclhip_decompress_track_batch takes requests 256 b .. 256 b + 255 and runs
    on XCD b % 8 -- a permutation; every XCD serves ONE contiguous range of the requests bucketed by clip (so the XCDs' clip sets overlap
    in at most the seven clips that straddle two ranges); requests of a clip keep their relative order"""
    rng = np.random.default_rng(num_requests)
    clips = rng.integers(3, 3 + 200, size=num_requests).astype(np.uint32)
    order = runtime.order_track_requests_for_locality(clips)
    assert np.array_equal(np.sort(order), np.arange(num_requests))
    ordered = clips[order]
    xcd_of = (np.arange(num_requests) // 256) % 8
    served = [ordered[xcd_of == x] for x in range(8)]
    for x in range(8):
        assert np.all(np.diff(served[x].astype(np.int64)) >= 0)                     # bucketed by clip inside its range
        where = order[xcd_of == x]
        for clip in np.unique(served[x]):
            mine = where[served[x] == clip]
            assert np.all(np.diff(mine.astype(np.int64)) > 0)                       # stable
    for x in range(7):
        if served[x].size and served[x + 1].size:
            assert served[x].max() <= served[x + 1].min()                           # the ranges follow one another in the bucketed sequence
    distinct = sum(np.unique(s).size for s in served)
    assert distinct <= np.unique(clips).size + 7


def test_track_request_order_takes_any_handles_and_refuses_null_lists():
    clips = (np.random.default_rng(1).integers(0, 5, size=3000) * 900000007 % (1 << 32)).astype(np.uint32)      # (not small numbers: the comparison sort)
    order = runtime.order_track_requests_for_locality(clips)
    assert np.array_equal(np.sort(order), np.arange(clips.size))
    assert np.array_equal(np.sort(clips[order][: 256]), clips[order][: 256])
    lib = runtime.load_library()
    assert lib.aclhip_order_track_requests_for_locality(None, 5, None) != 0
    assert lib.aclhip_order_track_requests_for_locality(None, 0, None) == 0
