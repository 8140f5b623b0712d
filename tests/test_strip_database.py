"""aclhip_strip_database_tier (host only, no GPU) against strip_database_quality_tier of the reference
(compression/compress.h:124): byte for byte the same compressed_database for the inline and the split form of every database
fixture and both tiers (tests/golden/database/*.npz hold the reference's outputs, make_golden_database.py), and live against
oracle/_ref where the reference was built."""
import ctypes

import numpy as np
import pytest

import helpers
from acl_amd import runtime, synth
from oracle import bindings as ob

FORMS = [("inline", "database_inline"), ("split", "database")]


@pytest.mark.parametrize("name", helpers.database_golden_cases())
def test_stripped_databases_match_the_reference_byte_for_byte(name):
    case = helpers.load_database_golden(name)
    for form, key in FORMS:
        for tier, tier_name in ((1, "medium"), (2, "low")):
            expected = case[f"stripped_{form}_{tier_name}"]
            status, stripped = runtime.strip_database_tier(case[key], tier)
            if expected.size == 0:
                assert status == runtime.ERROR_INVALID_ARGUMENT and stripped is None        # "Cannot strip an empty quality tier"
                continue
            assert status == runtime.OK
            assert np.array_equal(stripped, expected), (name, form, tier_name)
            # what is left is a valid database (inline: self contained; split: the kept tier's bulk data is passed in)
            kept = "bulk_low" if tier == 1 else "bulk_medium"
            medium, low = (None, case[kept]) if tier == 1 else (case[kept], None)
            if form == "inline":
                medium = low = None
            if form == "inline" or case[kept].size:
                check_status, message = runtime.check_database(stripped, medium if medium is not None and medium.size else None, low if low is not None and low.size else None)
                assert check_status == runtime.OK, message


def test_refusals():
    case = helpers.load_database_golden("two_clips_single_chunk")
    database = case["database_inline"]
    assert runtime.strip_database_tier(database, 0)[0] == runtime.ERROR_INVALID_ARGUMENT       # the high importance tier lives in the clips
    assert runtime.strip_database_tier(database, 3)[0] == runtime.ERROR_INVALID_ARGUMENT
    corrupt = synth.aligned_bytes(database.size)
    corrupt[:] = database
    corrupt[100] ^= 0x40                                                                       # is_valid(true): hash mismatch
    assert runtime.strip_database_tier(corrupt, 1)[0] == runtime.ERROR_INVALID_CLIP
    assert runtime.strip_database_tier(database[:40], 1)[0] == runtime.ERROR_INVALID_CLIP
    # stripping twice: the second tier can still go, the first cannot go again
    status, once = runtime.strip_database_tier(database, 1)
    assert status == runtime.OK
    assert runtime.strip_database_tier(once, 1)[0] == runtime.ERROR_INVALID_ARGUMENT
    status, twice = runtime.strip_database_tier(once, 2)
    assert status == runtime.OK and twice.size < once.size


@pytest.mark.skipif(not ob.have_ref_database(), reason="oracle/_ref/libaclref_db.so not built (no /root/reference)")
@pytest.mark.parametrize("seed", range(3))
def test_live_against_the_reference(seed):
    rng = np.random.default_rng(seed)
    clips = []
    for i in range(int(rng.integers(1, 4))):
        raw = synth.build_clip(seed=700 + seed * 10 + i, num_tracks=int(rng.integers(5, 30)), num_samples=int(rng.integers(40, 120)), with_side_data=True)
        clips.append(ob.ref_db_compress(raw.raw_keyframes, raw.sample_rate))
    reference = ob.ReferenceDatabase(clips, medium_proportion=float(rng.uniform(0.1, 0.4)), low_proportion=float(rng.uniform(0.1, 0.4)), max_chunk_size=int(rng.choice([4096, 16384])))
    try:
        for split, database in ((False, reference.database_inline), (True, reference.database)):
            for tier in (1, 2):
                expected = reference.strip(tier, split)
                status, stripped = runtime.strip_database_tier(database, tier)
                if expected is None:
                    assert status != runtime.OK
                else:
                    assert status == runtime.OK and np.array_equal(stripped, expected)
    finally:
        reference.close()


def _fnv1a32(data):
    data = np.ascontiguousarray(data)
    return ob.oracle().aclo_hash32(data.ctypes.data, data.size)     # hash32 (core/hash.h:86-99)


@pytest.mark.parametrize("seed", range(2))
def test_corrupt_headers_with_valid_hashes_never_crash_the_stripper(seed):
    """is_valid(true) only proves the bytes are the ones that were hashed: counts and offsets of a re-hashed, corrupted header must be
    refused or handled, never followed out of the buffer"""
    rng = np.random.default_rng(200 + seed)
    survived = 0
    for name in helpers.database_golden_cases():
        case = helpers.load_database_golden(name)
        for key in ("database_inline", "database"):
            original = case[key]
            for _ in range(120):
                database = synth.aligned_bytes(original.size)
                database[:] = original
                for _ in range(int(rng.integers(1, 4))):
                    at = int(rng.integers(8, 64)) if rng.uniform() < 0.8 else int(rng.integers(8, original.size))     # the database_header sits at [8, 64)
                    database[at] = int(rng.integers(0, 256))
                size = original.size if rng.uniform() < 0.9 else int(rng.integers(64, original.size))
                if rng.uniform() < 0.8:
                    database[0:4] = np.frombuffer(np.uint32(size).tobytes(), dtype=np.uint8)
                    database[4:8] = np.frombuffer(np.uint32(_fnv1a32(database[8:size])).tobytes(), dtype=np.uint8)
                for tier in (1, 2):
                    # (a corrupt bulk data size can ask for gigabytes, legitimately: only build what stays small)
                    view = database[:size]
                    needed = ctypes.c_uint64(0)
                    status = runtime.load_library().aclhip_strip_database_tier(view.ctypes.data, view.size, tier, None, 0, ctypes.byref(needed))
                    if status != runtime.OK or needed.value > (4 << 20):
                        continue
                    status, stripped = runtime.strip_database_tier(view, tier)
                    if status == runtime.OK:
                        survived += 1
                        assert stripped.size >= 64 and int(np.frombuffer(stripped[:4].tobytes(), dtype=np.uint32)[0]) == stripped.size
    assert survived > 0         # some corruptions only touch fields the stripper copies through
