"""The CPU restatement of database_context (oracle/database.py) + seek through the tier metadata (oracle/acl_oracle.c) against
golden vectors from the reference's own database pipeline (tests/golden/database/*.npz, see make_golden_database.py). No GPU."""
import numpy as np
import pytest

from oracle import bindings as ob
from oracle.database import OracleDatabase
import helpers


def _check_state(case, database, state, label):
    for c, clip in enumerate(case["clips"]):
        num_tracks = ob.oracle().aclo_num_tracks(clip.ctypes.data)
        for p, policy in enumerate(case["policies"]):
            for i, t in enumerate(case["times"][c]):
                out = database.decompress_tracks(clip, float(t), int(policy))
                expected = case["poses"][state, c, p, i, :num_tracks]
                assert helpers.bit_equal(out, expected), f"{label}: state {state} clip {c} policy {policy} time {t}: {helpers.max_abs_diff(out, expected)}"


@pytest.mark.parametrize("name", helpers.database_golden_cases())
@pytest.mark.parametrize("inline", [False, True])
def test_oracle_database_streaming_matches_reference_golden(name, inline):
    case = helpers.load_database_golden(name)
    database = OracleDatabase(case["database_inline"]) if inline else OracleDatabase(case["database"], case["bulk_medium"], case["bulk_low"])
    assert all(database.contains(clip) for clip in case["clips"])
    _check_state(case, database, 0, name)
    for state, (tier, num_chunks, stream_in) in enumerate(case["ops"]):
        moved = database.stream_in(int(tier), int(num_chunks)) if stream_in else database.stream_out(int(tier), int(num_chunks))
        # database_stream_request_result: 0 = done (nothing to do), 1 = dispatched
        assert (moved != 0) == (case["results"][state] == 1)
        _check_state(case, database, state + 1, name)


def test_fully_streamed_database_reproduces_unsplit_quality():
    """With every tier resident no keyframe is missing: interpolation happens between adjacent samples again, so poses at
    sample times equal what stream-in state 'all' gives with rounding none and nearest alike."""
    case = helpers.load_database_golden("three_clips_4k_chunks")
    database = OracleDatabase(case["database"], case["bulk_medium"], case["bulk_low"])
    database.stream_in(1)
    database.stream_in(2)
    assert database.is_streamed_in(1) and database.is_streamed_in(2)
    clip = case["clips"][0]
    for sample in (0, 7, 33, 100):
        t = sample / 30.0
        assert helpers.bit_equal(database.decompress_tracks(clip, t, ob.ROUND_NONE), database.decompress_tracks(clip, t, ob.ROUND_NEAREST))


def test_oracle_database_rejects_foreign_clips():
    from acl_amd import synth
    case = helpers.load_database_golden("two_clips_single_chunk")
    database = OracleDatabase(case["database"], case["bulk_medium"], case["bulk_low"])
    assert not database.contains(synth.build_clip(seed=5, num_tracks=8, num_samples=20).blob)
