"""The kernels' correctly rounded square root and reciprocal (acl_amd/csrc/aclhip_device.h: sqrt_rn, rcp_rn -- shorter instruction
sequences than the compiler's sqrtf / 1.0f / x) against the compiler's on EVERY float bit pattern, on the device: what bit exactness
with the reference's quat_from_positive_w / quat_normalize (includes/acl/math/quatf.h:135-211) rests on. Needs a GPU."""
import pytest

from acl_amd import runtime

pytestmark = pytest.mark.gpu


def test_short_sqrt_and_reciprocal_are_the_compilers_bit_for_bit_on_all_floats():
    with runtime.Context(0) as context:
        assert context.selftest_exact_math() == (0, 0, 0, 0)
