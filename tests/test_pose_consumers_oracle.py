"""Pose consumers, CPU side: the oracle's restatement of apply_additive_to_base (core/additive_utils.h:150) and
local_to_object_space (compression/transform_pose_utils.h:35) against the reference's own functions -- live when oracle/_ref was
built here, and through the committed fixtures (tests/golden/consumers, made by make_golden_consumers.py) everywhere.

Tolerances. Additive apply is bit exact (quat_mul restated in RTM's SSE2 summation order). Object space goes through
rtm::qvv_normalize, whose x86 form starts from the hardware reciprocal square root ESTIMATE: the reference itself is not
reproducible between CPU vendors there; the restatement normalizes with the decoder's correctly rounded sqrt + division and
agrees to a few ulp per level of the hierarchy -- 1e-5 (north_star's tolerance) on rotations, 1e-5 relative to the pose's extent
on translations."""
import numpy as np
import pytest

import helpers
from oracle import bindings as ob

ROTATION_TOLERANCE = 1.0e-5
TRANSLATION_TOLERANCE = 1.0e-5      # relative to max(1, largest |translation| of the pose)


def assert_object_space_close(actual, expected):
    assert np.abs(actual[..., 0:4] - expected[..., 0:4]).max() <= ROTATION_TOLERANCE
    extent = max(1.0, float(np.abs(expected[..., 4:7]).max()))
    assert np.abs(actual[..., 4:7] - expected[..., 4:7]).max() <= TRANSLATION_TOLERANCE * extent
    assert helpers.exact(actual[..., 8:11], expected[..., 8:11])        # scales are plain products: exact


def oracle_pipeline(case, index, additive_format):
    """oracle decode of both clips -> oracle additive apply -> oracle object space"""
    additive_pose = ob.oracle_decompress_tracks(case["additive_blob"], float(case["times"][index, 0]))
    base_pose = ob.oracle_decompress_tracks(case["base_blob"], float(case["times"][index, 1]))
    local = ob.oracle_apply_additive_to_base(additive_format, base_pose, additive_pose)
    return local, ob.oracle_local_to_object_space(case["parents"], local)


@pytest.mark.parametrize("name", helpers.consumer_golden_cases())
def test_oracle_matches_reference_fixtures(name):
    case = helpers.load_consumer_golden(name)
    for additive_format in range(4):
        for index in range(case["times"].shape[0]):
            local, object_space = oracle_pipeline(case, index, additive_format)
            assert helpers.bit_equal(local, case["local"][additive_format, index]), (name, additive_format, index)
            assert_object_space_close(object_space, case["object_space"][additive_format, index])


def random_pose(rng, num_transforms, scale_range=(0.8, 1.25)):
    pose = np.zeros((num_transforms, 12), dtype=np.float32)
    rotation = rng.normal(size=(num_transforms, 4)).astype(np.float32)
    pose[:, 0:4] = rotation / np.linalg.norm(rotation, axis=1, keepdims=True)
    pose[:, 4:7] = rng.normal(size=(num_transforms, 3)) * 0.5
    pose[:, 8:11] = rng.uniform(*scale_range, size=(num_transforms, 3))
    return pose


@pytest.mark.skipif(not ob.have_ref_pose(), reason="oracle/_ref/libaclref_pose.so not built (no /root/reference)")
@pytest.mark.parametrize("seed", range(8))
def test_additive_apply_is_bit_exact_against_the_reference(seed):
    rng = np.random.default_rng(seed)
    num_transforms = int(rng.integers(1, 300))
    base, additive = random_pose(rng, num_transforms), random_pose(rng, num_transforms, scale_range=(-0.2, 0.2) if seed % 2 else (0.8, 1.25))
    for additive_format in range(4):
        ours = ob.oracle_apply_additive_to_base(additive_format, base, additive)
        theirs = ob.ref_apply_additive_to_base(additive_format, base, additive)
        assert helpers.bit_equal(ours, theirs), additive_format


@pytest.mark.skipif(not ob.have_ref_pose(), reason="oracle/_ref/libaclref_pose.so not built (no /root/reference)")
@pytest.mark.parametrize("num_transforms,parent_span", [(1, 1), (2, 1), (70, 5), (100, 100), (320, 3), (1200, 30)])
def test_object_space_against_the_reference(num_transforms, parent_span):
    rng = np.random.default_rng(num_transforms)
    parents = np.array([0] + [rng.integers(max(0, i - parent_span), i) for i in range(1, num_transforms)], dtype=np.uint32)
    local = random_pose(rng, num_transforms, scale_range=(0.95, 1.05))
    assert_object_space_close(ob.oracle_local_to_object_space(parents, local), zero_w(ob.ref_local_to_object_space(parents, local)))


def zero_w(pose):
    pose = pose.copy()
    pose[:, 7] = 0.0
    pose[:, 11] = 0.0
    return pose


def test_quat_mul_summation_order():
    """(a*rw + b*rx) + (c*ry + d*rz) per lane: differs from the left-to-right sum in the last bit for some inputs, and the oracle
    must follow the pairwise form"""
    rng = np.random.default_rng(5)
    differs = 0
    for _ in range(200):
        lhs, rhs = rng.normal(size=4).astype(np.float32), rng.normal(size=4).astype(np.float32)
        lx, ly, lz, lw = lhs
        rx, ry, rz, rw = rhs
        pairwise = np.array([(rw * lx + rx * lw) + (ry * lz - rz * ly), (rw * ly - rx * lz) + (ry * lw + rz * lx),
                             (rw * lz + rx * ly) + (-(ry * lx) + rz * lw), (rw * lw - rx * lx) + (-(ry * ly) - rz * lz)], dtype=np.float32)
        sequential = np.array([rw * lx + rx * lw + ry * lz - rz * ly, rw * ly - rx * lz + ry * lw + rz * lx,
                               rw * lz + rx * ly - ry * lx + rz * lw, rw * lw - rx * lx - ry * ly - rz * lz], dtype=np.float32)
        assert helpers.exact(ob.oracle_quat_mul(lhs, rhs), pairwise)
        differs += int(not helpers.exact(pairwise, sequential))
    assert differs > 0


def test_object_space_roots_and_aliasing():
    rng = np.random.default_rng(9)
    local = random_pose(rng, 6)
    parents = np.array([ob.INVALID_PARENT, 0, ob.INVALID_PARENT, 2, 1, 3], dtype=np.uint32)
    out = ob.oracle_local_to_object_space(parents, local)
    assert helpers.exact(out[0], local[0]) and helpers.exact(out[2], local[2])     # roots pass through
    # a second root's subtree is what the single-root walk makes of it on its own
    sub = ob.oracle_local_to_object_space(np.array([0, 0, 1], dtype=np.uint32), local[[2, 3, 5]])
    assert helpers.exact(out[[2, 3, 5]], sub)
