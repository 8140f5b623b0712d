"""Known-answer tests restated from the reference's own unit tests, run against the CPU oracle (no GPU).

Vectors: /root/reference/tests/sources/core/test_interpolation_utils.cpp:33-409 and
/root/reference/tests/sources/math/test_vector4_packing.cpp:359-465 (loops and constants, the reference ships no files)."""
import ctypes

import numpy as np
import pytest

from oracle import bindings as ob

NONE, FLOOR, CEIL, NEAREST, PER_TRACK = 0, 1, 2, 3, 4
CLAMP, WRAP = 0, 1
THRESHOLD = 1.0e-6
F = np.float32


def _with_duration(num_samples, duration, t, policy, looping):
    k0, k1, alpha = ctypes.c_uint32(), ctypes.c_uint32(), ctypes.c_float()
    ob.oracle().aclo_find_linear_interpolation_samples_with_duration(num_samples, ctypes.c_float(duration), ctypes.c_float(t), policy, looping, ctypes.byref(k0), ctypes.byref(k1), ctypes.byref(alpha))
    return k0.value, k1.value, alpha.value


def _with_rate(num_samples, rate, t, policy, looping):
    k0, k1, alpha = ctypes.c_uint32(), ctypes.c_uint32(), ctypes.c_float()
    ob.oracle().aclo_find_linear_interpolation_samples_with_sample_rate(num_samples, ctypes.c_float(rate), ctypes.c_float(t), policy, looping, ctypes.byref(k0), ctypes.byref(k1), ctypes.byref(alpha))
    return k0.value, k1.value, alpha.value


def _t(x):
    return float(F(x) / F(30.0))


# (num_samples, duration, time, policy, looping) -> (key0, key1, alpha or tuple of acceptable alphas)
DURATION_VECTORS = [
    # clamp (test_interpolation_utils.cpp:37-113)
    ((31, 1.0, 0.0, NONE, CLAMP), (0, 1, 0.0)),
    ((31, 1.0, _t(1.0), NONE, CLAMP), (1, 2, 0.0)),
    ((31, 1.0, _t(2.5), NONE, CLAMP), (2, 3, 0.5)),
    ((31, 1.0, 1.0, NONE, CLAMP), (30, 30, 0.0)),
    ((31, 1.0, _t(2.5), FLOOR, CLAMP), (2, 3, 0.0)),
    ((31, 1.0, _t(2.5), CEIL, CLAMP), (2, 3, 1.0)),
    ((31, 1.0, _t(2.4), NEAREST, CLAMP), (2, 3, 0.0)),
    ((31, 1.0, _t(2.6), NEAREST, CLAMP), (2, 3, 1.0)),
    ((1, 0.0, 0.0, NONE, CLAMP), (0, 0, 0.0)),
    ((1, 0.0, 0.0, FLOOR, CLAMP), (0, 0, 0.0)),
    ((1, 0.0, 0.0, CEIL, CLAMP), (0, 0, 1.0)),
    ((1, 0.0, 0.0, NEAREST, CLAMP), (0, 0, 0.0)),
    # wrap (:115-210)
    ((30, 1.0, 0.0, NONE, WRAP), (0, 1, 0.0)),
    ((30, 1.0, _t(1.0), NONE, WRAP), (1, 2, 0.0)),
    ((30, 1.0, _t(2.5), NONE, WRAP), (2, 3, 0.5)),
    ((30, 1.0, 1.0, NONE, WRAP), (0, 0, 0.0)),
    ((30, 1.0, _t(2.5), FLOOR, WRAP), (2, 3, 0.0)),
    ((30, 1.0, _t(2.5), CEIL, WRAP), (2, 3, 1.0)),
    ((30, 1.0, _t(2.4), NEAREST, WRAP), (2, 3, 0.0)),
    ((30, 1.0, _t(2.6), NEAREST, WRAP), (2, 3, 1.0)),
    ((1, _t(1.0), 0.0, NONE, WRAP), (0, 0, 0.0)),
    ((1, _t(1.0), 0.0, FLOOR, WRAP), (0, 0, 0.0)),
    ((1, _t(1.0), 0.0, CEIL, WRAP), (0, 0, 1.0)),
    ((1, _t(1.0), 0.0, NEAREST, WRAP), (0, 0, 0.0)),
    ((1, _t(1.0), _t(1.0), NONE, WRAP), (0, 0, (0.0, 1.0))),
    ((1, _t(1.0), _t(0.5), NONE, WRAP), (0, 0, 0.5)),
    ((1, _t(1.0), _t(1.0), FLOOR, WRAP), (0, 0, 0.0)),
    ((1, _t(1.0), _t(1.0), CEIL, WRAP), (0, 0, 1.0)),
    ((1, _t(1.0), _t(1.0), NEAREST, WRAP), (0, 0, (0.0, 1.0))),
]

RATE_VECTORS = [
    # clamp (:214-262)
    ((31, 30.0, 0.0, NONE, CLAMP), (0, 1, 0.0)),
    ((31, 30.0, _t(1.0), NONE, CLAMP), (1, 2, 0.0)),
    ((31, 30.0, _t(2.5), NONE, CLAMP), (2, 3, 0.5)),
    ((31, 30.0, 1.0, NONE, CLAMP), (30, 30, 0.0)),
    ((31, 30.0, _t(2.5), FLOOR, CLAMP), (2, 3, 0.0)),
    ((31, 30.0, _t(2.5), CEIL, CLAMP), (2, 3, 1.0)),
    ((31, 30.0, _t(2.4), NEAREST, CLAMP), (2, 3, 0.0)),
    ((31, 30.0, _t(2.6), NEAREST, CLAMP), (2, 3, 1.0)),
    # wrap (:264-312)
    ((30, 30.0, 0.0, NONE, WRAP), (0, 1, 0.0)),
    ((30, 30.0, _t(1.0), NONE, WRAP), (1, 2, 0.0)),
    ((30, 30.0, _t(2.5), NONE, WRAP), (2, 3, 0.5)),
    ((30, 30.0, 1.0, NONE, WRAP), (0, 0, 0.0)),
    ((30, 30.0, _t(2.5), FLOOR, WRAP), (2, 3, 0.0)),
    ((30, 30.0, _t(2.5), CEIL, WRAP), (2, 3, 1.0)),
    ((30, 30.0, _t(2.4), NEAREST, WRAP), (2, 3, 0.0)),
    ((30, 30.0, _t(2.6), NEAREST, WRAP), (2, 3, 1.0)),
]


def _check(result, expected):
    k0, k1, alpha = result
    e0, e1, ealpha = expected
    assert (k0, k1) == (e0, e1)
    acceptable = ealpha if isinstance(ealpha, tuple) else (ealpha,)
    assert any(abs(alpha - a) < THRESHOLD for a in acceptable), (alpha, ealpha)


@pytest.mark.parametrize("args,expected", DURATION_VECTORS)
def test_find_linear_interpolation_samples_with_duration(args, expected):
    _check(_with_duration(*args), expected)


@pytest.mark.parametrize("args,expected", RATE_VECTORS)
def test_find_linear_interpolation_samples_with_sample_rate(args, expected):
    _check(_with_rate(*args), expected)


ALPHA_VECTORS = [
    # (sample_index, index0, index1) -> alpha per policy none / floor / ceil / nearest (:316-374)
    ((0.0, 1, 1), (0.0, 0.0, 1.0, 0.0)),
    ((1.5, 1, 2), (0.5, 0.0, 1.0, 1.0)),
    ((1.5, 0, 2), (0.75, 0.0, 1.0, 1.0)),
    ((1.5, 0, 3), (0.5, 0.0, 1.0, 1.0)),
    ((1.5, 1, 4), (0.16666667, 0.0, 1.0, 0.0)),
]


@pytest.mark.parametrize("looping", [CLAMP, WRAP])
@pytest.mark.parametrize("args,expected", ALPHA_VECTORS)
def test_find_linear_interpolation_alpha(args, expected, looping):
    for policy, value in zip((NONE, FLOOR, CEIL, NEAREST), expected):
        alpha = ob.oracle().aclo_find_linear_interpolation_alpha(ctypes.c_float(args[0]), args[1], args[2], policy, looping)
        assert abs(alpha - value) < THRESHOLD


def test_find_linear_interpolation_alpha_wrapping_back_to_the_first_sample():
    for policy, value in zip((NONE, FLOOR, CEIL, NEAREST), (0.5, 0.0, 1.0, 1.0)):
        assert abs(ob.oracle().aclo_find_linear_interpolation_alpha(ctypes.c_float(2.5), 2, 0, policy, WRAP) - value) < THRESHOLD


def test_apply_rounding_policy():
    f = ob.oracle().aclo_apply_rounding_policy
    for alpha, expected in ((0.2, (0.2, 0.0, 1.0, 0.0, 0.2)), (0.8, (0.8, 0.0, 1.0, 1.0, 0.8))):
        for policy, value in zip((NONE, FLOOR, CEIL, NEAREST, PER_TRACK), expected):
            assert abs(f(ctypes.c_float(alpha), policy) - value) < THRESHOLD


def test_pack_vector3_uXX_round_trip_exhaustive_low_rates():
    # every value of every width 1..19 at bit offsets {0,1,5,31,32,33,63,64,65,93} (test_vector4_packing.cpp:385-457, "part0")
    assert ob.oracle().aclo_selftest_pack_vector3_uXX(1, 19) == 0


def test_pack_vector3_uXX_round_trip_exhaustive_high_rates():
    # the rest of "part0" and "part1": widths 20..23
    assert ob.oracle().aclo_selftest_pack_vector3_uXX(20, 23) == 0


def test_unpack_vector3_u24_every_value():
    # pack_vector3_24 (test_vector4_packing.cpp:359-383): all 256 values
    out = np.zeros(3, dtype=np.float32)
    for value in range(256):
        data = np.array([value, 255 - value, value] + [0] * 13, dtype=np.uint8)
        ob.oracle().aclo_unpack_vector3_u24(data.ctypes.data, out.ctypes.data)
        assert abs(out[0] - min(F(value) / F(255.0), 1.0)) < THRESHOLD
        assert abs(out[1] - min(F(255 - value) / F(255.0), 1.0)) < THRESHOLD


def test_unpack_vector3_u48_and_96():
    out = np.zeros(3, dtype=np.float32)
    data = np.zeros(32, dtype=np.uint8)
    data[:6] = np.array([0x34, 0x12, 0xFF, 0xFF, 0x00, 0x80], dtype=np.uint8)       # little endian u16: 0x1234, 0xFFFF, 0x8000
    ob.oracle().aclo_unpack_vector3_u48(data.ctypes.data, out.ctypes.data)
    assert np.allclose(out, np.array([0x1234, 0xFFFF, 0x8000], dtype=np.float32) / F(65535.0), atol=THRESHOLD)

    values = np.array([1.5, -2.25, 1e-3], dtype=np.float32)
    big_endian = values.view(np.uint32).byteswap().view(np.uint8)
    for offset in (0, 1, 5, 31, 32, 33, 63, 64, 65, 93):
        buf = np.zeros(40, dtype=np.uint8)
        ob.oracle().aclo_memcpy_bits(buf.ctypes.data, offset, big_endian.ctypes.data, 0, 96)
        ob.oracle().aclo_unpack_vector3_96(buf.ctypes.data, offset, out.ctypes.data)
        assert np.array_equal(out.view(np.uint32), values.view(np.uint32))


def test_memcpy_bits():
    # core/memory_utils.h:295-335 as exercised by tests/sources/core/test_memory_utils.cpp:141
    src = np.array([0xA5, 0x5A, 0xFF, 0x00], dtype=np.uint8)
    dst = np.zeros(8, dtype=np.uint8)
    ob.oracle().aclo_memcpy_bits(dst.ctypes.data, 3, src.ctypes.data, 0, 16)
    bits = np.unpackbits(dst)
    assert np.array_equal(bits[3:19], np.unpackbits(src[:2]))
    assert bits[:3].sum() == 0 and bits[19:].sum() == 0
    dst[:] = 0xFF
    ob.oracle().aclo_memcpy_bits(dst.ctypes.data, 5, src.ctypes.data, 24, 8)      # copy zeros over ones
    bits = np.unpackbits(dst)
    assert bits[5:13].sum() == 0 and bits[:5].sum() == 5 and bits[13:].sum() == 51


def test_time_utils():
    # tests/sources/core/test_time_utils.cpp:35-57
    lib = ob.oracle()
    inf = float("inf")
    assert lib.aclo_calculate_num_samples(0.0, 30.0) == 0
    assert lib.aclo_calculate_num_samples(1.0, 30.0) == 31
    assert lib.aclo_calculate_num_samples(1.0, 24.0) == 25
    assert lib.aclo_calculate_num_samples(F(1.0) / F(30.0), 30.0) == 2
    assert lib.aclo_calculate_num_samples(inf, 30.0) == 1
    assert lib.aclo_calculate_duration(0, 30.0) == 0.0
    assert lib.aclo_calculate_duration(1, 30.0) == inf
    assert lib.aclo_calculate_duration(1, 8.0) == inf
    assert abs(lib.aclo_calculate_duration(31, 30.0) - 1.0) < 1.0e-8
    assert abs(lib.aclo_calculate_duration(9, 8.0) - 1.0) < 1.0e-8
    assert lib.aclo_calculate_finite_duration(0, 30.0) == 0.0
    assert lib.aclo_calculate_finite_duration(1, 30.0) == 0.0
    assert lib.aclo_calculate_finite_duration(1, 8.0) == 0.0
    assert abs(lib.aclo_calculate_finite_duration(31, 30.0) - 1.0) < 1.0e-8
    assert abs(lib.aclo_calculate_finite_duration(9, 8.0) - 1.0) < 1.0e-8


def test_bit_manip_utils():
    # tests/sources/core/test_bit_manip_utils.cpp:31-60, the 32 bit forms the database seek uses
    lib = ob.oracle()
    for value, expected in ((0x00000000, 0), (0x00000001, 1), (0x10000000, 1), (0x10101001, 4), (0xFFFFFFFF, 32)):
        assert lib.aclo_count_set_bits(value) == expected
    for value, expected in ((0x00000000, 32), (0x00000001, 31), (0x00000002, 30), (0x80000000, 0), (0x40000000, 1)):
        assert lib.aclo_count_leading_zeros(value) == expected
    for value, expected in ((0x00000000, 32), (0x00000001, 0), (0x00000002, 1), (0x80000000, 31), (0x40000000, 30)):
        assert lib.aclo_count_trailing_zeros(value) == expected


def test_scalar_packing_math():
    # tests/sources/math/test_scalar_packing.cpp:44-79: boundaries and the exhaustive unpack -> pack round trip for 1..22 bits
    assert ob.oracle().aclo_selftest_scalar_packing(1, 22) == 0


def test_unpack_scalarf_32_at_bit_offsets():
    # test_scalar_packing.cpp:81-117: a big endian fp32 at bit offsets {0,1,5,31,32,33,63,64,65,93}
    lib = ob.oracle()
    for value in (F(6123.123812), F(19237.01293127), F(0.913912387), F(-0.1816253)):
        big_endian = np.array([value], dtype=np.float32).view(np.uint32).byteswap().view(np.uint8)
        for offset in (0, 1, 5, 31, 32, 33, 63, 64, 65, 93):
            buf = np.zeros(32, dtype=np.uint8)
            lib.aclo_memcpy_bits(buf.ctypes.data, offset, big_endian.ctypes.data, 0, 32)
            assert F(lib.aclo_unpack_scalarf_32(buf.ctypes.data, offset)) == value


def test_unpack_scalarf_uXX_at_bit_offsets():
    # test_scalar_packing.cpp:119-166 (test_unpack_scalarf_uXX_unsafe): every value of widths 1..12 and samples of 13..23, written big
    # endian at the test's bit offsets, read back exactly as float(value) * (1 / max)
    lib = ob.oracle()
    rng = np.random.default_rng(0)
    for num_bits in range(1, 24):
        max_value = (1 << num_bits) - 1
        values = range(max_value + 1) if num_bits <= 8 else [0, 1, max_value - 1, max_value] + list(rng.integers(0, max_value + 1, size=60))
        for value in values:
            field = np.array([int(value) << (32 - num_bits)], dtype=np.uint32).byteswap().view(np.uint8)       # the field in the top bits, big endian
            for offset in (0, 1, 5, 31, 32, 33, 63, 64, 65, 93):
                buf = np.zeros(32, dtype=np.uint8)
                lib.aclo_memcpy_bits(buf.ctypes.data, offset, field.ctypes.data, 0, num_bits)
                expected = F(int(value)) * (F(1.0) / F(max_value))
                assert F(lib.aclo_unpack_scalarf_uXX(num_bits, buf.ctypes.data, offset)) == expected
                assert abs(lib.aclo_unpack_scalar_unsigned(int(value), num_bits) - expected) == 0.0
