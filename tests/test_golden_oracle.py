"""The CPU oracle against golden vectors produced by the reference's own decoder (tests/golden/*.npz, see make_golden.py). No GPU."""
import numpy as np
import pytest

from oracle import bindings as ob
import helpers


@pytest.mark.parametrize("name", helpers.golden_cases())
def test_oracle_decompress_tracks_matches_reference_golden(name):
    case = helpers.load_golden(name)
    options = helpers.oracle_options(case["settings"], case["default_mode"], case["defaults"], case["track_rounding"])
    for p, policy in enumerate(case["policies"]):
        for i, t in enumerate(case["times"]):
            out = case["prefill"].copy()
            ob.oracle_decompress_tracks(case["blob"], float(t), int(policy), options, out=out)
            expected = case["poses"][p, i]
            # whole pose path: the restatement follows the reference operation by operation -> bit exact
            assert helpers.bit_equal(out, expected), f"{name}: policy {policy} time {t}: max diff {helpers.max_abs_diff(out, expected)}"


@pytest.mark.parametrize("name", helpers.golden_cases())
def test_oracle_decompress_track_matches_reference_golden(name):
    case = helpers.load_golden(name)
    options = helpers.oracle_options(case["settings"], case["default_mode"], case["defaults"], case["track_rounding"])
    worst = 0.0
    for p, policy in enumerate(case["policies"]):
        for i, t in enumerate(case["times"]):
            track = int(case["track_indices"][i])
            out = case["prefill"][track].copy()
            result = ob.oracle().aclo_decompress_track(case["blob"].ctypes.data, float(t), int(policy), options, track, out.ctypes.data)
            assert result == 0
            worst = max(worst, helpers.max_abs_diff(out, case["single"][p, i]))
    # single track path: the reference goes through RTM's quat_lerp / quat_normalize (reciprocal square root estimate +
    # Newton-Raphson, hardware dependent in the last bits); the reference's own validator allows 1e-4 here
    # (tools/acl_compressor/sources/validate_tracks.cpp:41-46,112-118). We hold 1e-6.
    assert worst <= 1e-6, f"{name}: {worst}"
