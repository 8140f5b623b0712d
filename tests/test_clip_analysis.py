"""aclhip_analyze_clip (host only): the facts registration derives about the VALUES a clip can decode to -- above all whether the short
correctly rounded square root / reciprocal of aclhip_device.h may run on its rotations (k_clip_short_exact_math). The
claim is checked here against brute force on the clip's actual samples: every key frame of every animated rotation decoded by the
oracle without normalization (so x, y, z, W come back as the decoder built them), W^2 = |((1 - x^2) - y^2) - z^2| recomputed in fp32
one operation at a time (math/quatf.h:135-147) -- a clip that carries the fact has no such argument in (0, 2^-96), and clips built to
have one do not carry it. No GPU needed."""
import numpy as np
import pytest

from acl_amd import runtime, synth
from oracle import bindings as ob

GAP = np.float32(2.0 ** -96)


def _square_root_arguments(clip):
    """fp32 W^2 of every rotation of every key frame, the way the decoder computes it (quat_from_positive_w)."""
    options = ob.default_options(normalization=0)
    arguments = []
    for key in range(clip.spec.num_samples):
        time = min(key / clip.spec.sample_rate, clip.duration)
        pose = ob.oracle_decompress_tracks(clip.blob, float(time), 1, options)       # floor: the key frame itself
        x, y, z = pose[:, 0].astype(np.float32), pose[:, 1].astype(np.float32), pose[:, 2].astype(np.float32)
        one = np.float32(1.0)
        arguments.append(np.abs(((one - x * x) - y * y) - z * z).astype(np.float32))
    return np.concatenate(arguments)


SPECS = [
    dict(seed=2, num_tracks=100, num_samples=61, raw_fraction=0.0),
    dict(seed=3, num_tracks=37, num_samples=200, has_scale=1, scale_default=0.3, raw_fraction=0.0),
    dict(seed=4, num_tracks=300, num_samples=90, has_scale=1, scale_default=0.7, scale_constant=0.1, rotation_constant=0.45, translation_constant=0.8),   # the bench rig's mix: 1 % raw
    dict(seed=5, num_tracks=64, num_samples=50, strip_keyframes=1, wrap=1, raw_fraction=0.0),
    dict(seed=6, num_tracks=20, num_samples=33, raw_fraction=0.5, rotation_constant=0.1),
    dict(seed=7, num_tracks=50, num_samples=40, width0_fraction=0.5, raw_fraction=0.0),
]


@pytest.mark.parametrize("spec", SPECS, ids=lambda s: f"seed{s['seed']}")
def test_ordinary_clips_take_the_short_forms_and_none_of_their_samples_is_in_the_gap(spec):
    clip = synth.build_clip(**spec)
    facts = runtime.analyze_clip(clip.blob)
    assert facts & runtime.CLIP_FACT_SHORT_EXACT_MATH, "the analysis refuses an ordinary clip: the short forms would never run"
    arguments = _square_root_arguments(clip)
    assert not np.any((arguments > 0) & (arguments < GAP))
    # raw (fp32) rotation samples do not disqualify a clip: the waves that meet one take the compiler's forms for that pass
    if spec.get("raw_fraction") == 0.0:
        assert not facts & (runtime.CLIP_FACT_RAW_ROTATIONS | runtime.CLIP_FACT_NEGATIVE_SCALE)
    if spec.get("raw_fraction") == 0.5:
        assert facts & runtime.CLIP_FACT_RAW_ROTATIONS


def _patched(clip, rotation, minimum, extent):
    """The clip with the clip range of its `rotation`-th animated rotation replaced (core/impl/compressed_headers.h:227-263: counts,
    then offsets relative to the transform_tracks_header at +32; write_range_data.h:79-207: groups of <= 4 as min.x[g] min.y[g] .. ext.z[g])."""
    blob = clip.blob.copy()
    header = np.frombuffer(blob[32:32 + 52].tobytes(), dtype=np.uint32)
    num_animated_rotations, clip_range_offset = int(header[2]), int(header[12])
    assert rotation < num_animated_rotations
    group_index, lane = divmod(rotation, 4)
    group = min(4, num_animated_rotations - group_index * 4)
    base = 32 + clip_range_offset + group_index * 4 * 24
    values = blob[base: base + 6 * group * 4].view(np.float32)
    for c in range(3):
        values[c * group + lane] = minimum[c]
        values[(3 + c) * group + lane] = extent[c]
    aligned = synth.aligned_bytes(blob.size)
    aligned[:] = blob
    return aligned


def test_clips_that_can_reach_the_gap_are_refused():
    clip = synth.build_clip(seed=77, num_tracks=12, num_samples=20, rotation_default=0.0, rotation_constant=0.0, raw_fraction=0.0, width0_fraction=0.0)
    assert runtime.analyze_clip(clip.blob) & runtime.CLIP_FACT_SHORT_EXACT_MATH
    cases = {
        "x = 1 next to y = 1e-20: W^2 = 1e-40": ((1.0, 1.0e-20, 0.0), (0.0, 0.0, 0.0)),
        "x = 1 next to z = 1e-16 behind y = 0: W^2 = 1e-32": ((1.0, 0.0, 1.0e-16), (0.0, 0.0, 0.0)),
        "a negative extent (the decoded values are no longer ordered)": ((0.5, 0.1, 0.1), (-0.25, 0.1, 0.1)),
        "a range that is not a number": ((np.nan, 0.0, 0.0), (0.1, 0.1, 0.1)),
        "an infinite extent": ((0.0, 0.0, 0.0), (np.inf, 0.1, 0.1)),
        "values beyond 2^20": ((3.0e6, 0.0, 0.0), (1.0, 0.1, 0.1)),
    }
    for name, (minimum, extent) in cases.items():
        patched = _patched(clip, 1, minimum, extent)
        status, message = runtime.check_clip(patched, check_hash=False)
        assert status == 0, (name, message)
        facts = runtime.analyze_clip(patched, check_hash=False)
        assert not facts & runtime.CLIP_FACT_SHORT_EXACT_MATH, name
    # ... while ranges that stay away from zero, or sit exactly on it, keep the fact -- and so do ranges whose GRID holds values next
    # to zero as long as no stored key frame turns one into a square root argument in the gap (the second, exact test of registration)
    for name, (minimum, extent) in {"exact zeros": ((0.0, 0.0, 0.0), (0.0, 0.0, 0.0)), "small but not tiny": ((1.0e-6, -1.0e-6, 0.3), (0.0, 0.0, 0.1)),
                                    "a range that crosses zero in steps of 6e-15 and less": ((-1.0e-13, 0.0, 0.0), (2.0e-13, 0.5, 0.5)),
                                    "a tiny negative value next to ordinary ones": ((0.1, -1.0e-15, 0.2), (0.0, 0.0, 0.0)),
                                    "the noise of a hinge joint's idle axes: +- 1e-7 in steps of 1e-11 or so": ((-1.0000001e-7, -1.0000003e-7, -0.5), (2.0e-7, 2.0e-7, 1.0))}.items():
        facts = runtime.analyze_clip(_patched(clip, 1, minimum, extent), check_hash=False)
        assert facts & runtime.CLIP_FACT_SHORT_EXACT_MATH, name


@pytest.mark.parametrize("num_samples", [20, 100])
def test_the_exact_test_agrees_with_brute_force_on_the_stored_key_frames(num_samples):
    """Grids with values next to zero send registration to its second test: W^2 of every stored key frame. Same verdict as the oracle's
    decode of every key frame (several segments at 100 samples)."""
    clip = synth.build_clip(seed=78, num_tracks=9, num_samples=num_samples, rotation_default=0.0, rotation_constant=0.0, raw_fraction=0.0, width0_fraction=0.0)
    one_below = float(np.nextafter(np.float32(1.0), np.float32(0.0)))
    for minimum, extent in (((1.0, 0.0, 0.0), (0.0, 0.0, 1.0e-15)),            # z in [0, 1e-15] behind an exact cancellation: in the gap unless z == 0
                            ((1.0, 0.0, 0.0), (0.0, 0.0, 0.0)),                # W^2 == 0
                            ((one_below, 0.0, 0.0), (0.0, 0.0, 1.0e-15)),      # no cancellation: W^2 = 1.2e-7
                            ((0.6, 0.8, 0.0), (0.0, 0.0, 1.0e-15)),            # fl(1 - 0.36) - 0.64 does not cancel exactly in fp32
                            ((-1.0e-13, -1.0e-13, -1.0e-13), (2.0e-13, 2.0e-13, 2.0e-13))):
        patched = _patched(clip, 2, minimum, extent)
        facts = runtime.analyze_clip(patched, check_hash=False)

        class Patched:
            blob, spec, duration = patched, clip.spec, clip.duration
        arguments = _square_root_arguments(Patched)
        in_gap = bool(np.any((arguments > 0) & (arguments < GAP)))
        assert bool(facts & runtime.CLIP_FACT_SHORT_EXACT_MATH) == (not in_gap), (minimum, extent, in_gap)


def test_negative_scales_and_scalar_lists():
    mirrored = synth.build_clip(seed=9, num_tracks=10, num_samples=12, has_scale=1, scale_default=0.0, scale_constant=0.5, mirrored_scale_fraction=0.5)
    assert runtime.analyze_clip(mirrored.blob) & runtime.CLIP_FACT_NEGATIVE_SCALE
    curves = synth.build_scalar_clip(seed=1, track_type=0, num_tracks=8, num_samples=10)
    assert runtime.analyze_clip(curves.blob) == 0
    with pytest.raises(runtime.AclHipError):
        runtime.analyze_clip(np.zeros(64, dtype=np.uint8))
