"""aclhip_order_instances_for_locality (host only, no GPU): the decode order it gives is a permutation that keeps every clip on one
XCD -- workgroup b of a launch runs on XCD b % 8 and holds 4 consecutive instances -- and next to its other instances."""
import numpy as np
import pytest

from acl_amd import runtime


@pytest.mark.parametrize("num_instances,num_clips", [(0, 1), (1, 1), (3, 2), (4, 1), (1000, 7), (65536, 256), (10001, 1000), (4096, 8)])
def test_order_is_a_locality_preserving_permutation(num_instances, num_clips):
    rng = np.random.default_rng(num_instances + num_clips)
    clips = rng.integers(0, num_clips, size=num_instances).astype(np.uint32)
    order = runtime.order_instances_for_locality(clips)
    assert np.array_equal(np.sort(order), np.arange(num_instances))

    ordered = clips[order]
    whole = (num_instances // 4) * 4
    workgroups = ordered[:whole].reshape(-1, 4)
    xcd = np.arange(workgroups.shape[0]) % 8
    affine = (workgroups % 8 == xcd[:, None]).all(axis=1)
    if num_clips >= 64 and num_instances >= 4096:
        assert affine.mean() > 0.97          # evenly spread clips: only the tail is dealt to foreign XCDs
    # an XCD sees its clips in ascending order while its own list lasts: every clip's instances are neighbours
    for x in range(8):
        mine = workgroups[(xcd == x) & affine].reshape(-1)
        assert np.all(np.diff(mine.astype(np.int64)) >= 0)
    # stable: instances of one clip keep their relative order inside a workgroup sequence of their XCD
    for clip in np.unique(clips)[:16]:
        positions = order[ordered == clip]
        home = positions[: max(1, positions.size // 2)]
        assert np.all(np.diff(home.astype(np.int64)) > 0)


def test_skewed_batches_stay_balanced():
    """90 % of the instances on one clip: its XCD's list is dealt to every XCD once the others run dry, nothing is dropped"""
    rng = np.random.default_rng(1)
    clips = np.where(rng.uniform(size=20000) < 0.9, 5, rng.integers(0, 64, size=20000)).astype(np.uint32)
    order = runtime.order_instances_for_locality(clips)
    assert np.array_equal(np.sort(order), np.arange(clips.size))


def test_null_arguments():
    lib = runtime.load_library()
    assert lib.aclhip_order_instances_for_locality(None, None, 4, None) == runtime.ERROR_INVALID_ARGUMENT
    assert lib.aclhip_order_instances_for_locality(None, None, 0, None) == runtime.OK
