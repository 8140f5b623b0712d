"""aclhip_plan_hierarchy_walk (host only, no GPU): the schedule the pose consumers follow when they turn a local pose into an object
space pose (local_to_object_space, compression/transform_pose_utils.h:35-50, needs parents first). Valid for every forest, never
more than P transforms per step, and as short as a schedule can be (Hu's algorithm) -- checked against exhaustive search on small
forests and against the closed form max_k (k + ceil(#{transforms deeper than k} / P)) on large ones."""
import itertools

import numpy as np
import pytest

from acl_amd import runtime, synth

NO_PARENT = runtime.NO_PARENT


def random_forest(rng, num_tracks, parent_span, extra_roots=0):
    parents = np.zeros(num_tracks, dtype=np.uint32)
    parents[0] = NO_PARENT
    for i in range(1, num_tracks):
        parents[i] = rng.integers(max(0, i - parent_span), i)
    if extra_roots and num_tracks > 1:
        parents[rng.choice(np.arange(1, num_tracks), size=min(extra_roots, num_tracks - 1), replace=False)] = NO_PARENT
    return parents


def depths(parents):
    depth = np.zeros(parents.size, dtype=np.int64)
    for i in range(1, parents.size):
        depth[i] = 0 if parents[i] == NO_PARENT else depth[parents[i]] + 1
    return depth


def closed_form_steps(parents, per_step):
    depth = depths(parents)
    if depth.max(initial=0) == 0:
        return 0
    return max(k + -(-int((depth > k).sum()) // per_step) for k in range(0, int(depth.max()) + 1))


def check_schedule(parents, per_step):
    num_steps, steps = runtime.plan_hierarchy_walk(parents, per_step)
    depth = depths(parents)
    assert np.all((steps == 0) == (depth == 0))                               # roots are not walked, everything else is
    walked = np.flatnonzero(steps)
    assert np.all(steps[walked] > np.where(depth[parents[walked]] == 0, 0, steps[parents[walked]]))     # after its parent
    if walked.size:
        assert np.bincount(steps[walked]).max() <= per_step
        assert steps.max() == num_steps
    return num_steps


@pytest.mark.parametrize("per_step", [1, 2, 8, 16, 64])
def test_schedules_are_valid_and_match_the_closed_form(per_step):
    rng = np.random.default_rng(per_step)
    for num_tracks, span, roots in ((1, 1, 0), (2, 1, 0), (17, 1, 0), (100, 8, 0), (100, 100, 3), (400, 3, 1), (1200, 40, 5), (3000, 2, 0)):
        parents = random_forest(rng, num_tracks, span, roots)
        assert check_schedule(parents, per_step) == closed_form_steps(parents, per_step), (num_tracks, span, roots)


def test_humanoid_takes_one_step_per_depth():
    parents = synth.humanoid_hierarchy(100)
    assert int(depths(parents).max()) == 12
    assert check_schedule(parents, 64) == 12
    assert check_schedule(parents, 16) == 12
    assert check_schedule(parents, 8) == 14 == closed_form_steps(parents, 8)      # depth by depth, 8 at a time, it would be 19
    by_depth = sum(-(-int(count) // 8) for count in np.bincount(depths(parents))[1:])
    assert by_depth == 19


def exhaustive_minimum(parents, per_step):
    """fewest steps over ALL valid schedules (breadth first over sets of finished transforms)"""
    n = parents.size
    roots = frozenset(i for i in range(n) if i == 0 or parents[i] == NO_PARENT)
    everything = frozenset(range(n))
    frontier, steps = {roots}, 0
    while everything not in frontier:
        steps += 1
        following = set()
        for done in frontier:
            ready = [i for i in range(n) if i not in done and int(parents[i]) in done]
            take = min(per_step, len(ready))
            for chosen in itertools.combinations(ready, take):      # taking fewer than possible never helps
                following.add(done | frozenset(chosen))
        frontier = following
    return steps


@pytest.mark.parametrize("seed", range(12))
def test_schedules_are_optimal_on_small_forests(seed):
    rng = np.random.default_rng(100 + seed)
    num_tracks = int(rng.integers(2, 11))
    parents = random_forest(rng, num_tracks, int(rng.integers(1, 6)), int(rng.integers(0, 2)))
    for per_step in (1, 2, 3):
        assert check_schedule(parents, per_step) == exhaustive_minimum(parents, per_step), (parents, per_step)


def test_rejects_children_before_parents():
    lib = runtime.load_library()
    import ctypes
    num_steps = ctypes.c_uint32(0)
    for bad in (np.array([NO_PARENT, 2, 1], dtype=np.uint32), np.array([NO_PARENT, 1], dtype=np.uint32), np.array([NO_PARENT, 5], dtype=np.uint32)):
        assert lib.aclhip_plan_hierarchy_walk(bad.ctypes.data, bad.size, 8, None, ctypes.byref(num_steps)) == runtime.ERROR_INVALID_ARGUMENT
    ok = np.array([7, 0, NO_PARENT, 2], dtype=np.uint32)                # transform 0 is a root whatever its parent index says
    assert lib.aclhip_plan_hierarchy_walk(ok.ctypes.data, ok.size, 8, None, ctypes.byref(num_steps)) == runtime.OK and num_steps.value == 1
    assert lib.aclhip_plan_hierarchy_walk(ok.ctypes.data, ok.size, 0, None, ctypes.byref(num_steps)) == runtime.ERROR_INVALID_ARGUMENT
    assert lib.aclhip_plan_hierarchy_walk(None, 0, 8, None, ctypes.byref(num_steps)) == runtime.OK and num_steps.value == 0
