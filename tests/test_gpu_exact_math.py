"""The short correctly rounded square root / reciprocal of the rotation arithmetic (acl_amd/csrc/aclhip_device.h: sqrt_rn_short,
rcp_rn_short) are bit identical to sqrtf / 1.0f / x only for arguments that are 0 or >= 2^-96 (all 2^32 floats checked:
tools/probes/exact_math_probe.hip). Whether a clip's rotations can produce anything else is decided at registration
(host_clips.inl: k_clip_short_exact_math). Here: a clip built to hit the gap -- a rotation component of exactly 1 next to one of 1e-20:
W^2 = 1e-40 -- must be recognised (its poses stay bit identical to the oracle's, which computes with the C library's sqrtf), and the
same clip with the analysis overruled (ACLHIP_SHORT_EXACT_MATH=1, which only the lab build of the library -- libaclhip_lab.so,
-DACLHIP_LAB_KNOBS -- listens to: a shipped library cannot be talked out of bit exactness by an environment variable) must NOT be: the
analysis is what keeps the bits.
The reference's arithmetic: includes/acl/math/quatf.h:135-147,200-211. Needs a GPU."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_SCRIPT = r"""
import sys
import numpy as np
from acl_amd import runtime, synth
from oracle import bindings as ob

clip = synth.build_clip(seed=77, num_tracks=12, num_samples=20, rotation_default=0.0, rotation_constant=0.0, raw_fraction=0.0, width0_fraction=0.0)
blob = clip.blob.copy()
# transform_tracks_header at +32 (core/impl/compressed_headers.h:227-263): counts, then the offsets relative to the header itself
header = np.frombuffer(blob[32:32 + 52].tobytes(), dtype=np.uint32)
num_animated_rotations, clip_range_offset = int(header[2]), int(header[12])
group = min(4, num_animated_rotations)
base = 32 + clip_range_offset
values = blob[base: base + 6 * group * 4].view(np.float32)
# the first animated rotation: min = (1, 1e-20, 0), extent = 0: every key decodes to x = 1 exactly, y = 1e-20, z = 0
import os
values[0 * group], values[1 * group], values[2 * group] = float(os.environ.get("ACLHIP_TEST_X", "1.0")), float(os.environ.get("ACLHIP_TEST_TINY_COMPONENT", "1.0e-20")), 0.0
values[3 * group], values[4 * group], values[5 * group] = 0.0, 0.0, 0.0
aligned = synth.aligned_bytes(blob.size)
aligned[:] = blob

with runtime.Context(0) as context:
    handle = context.register_clip(aligned, check_hash=False)
    times = np.linspace(0.0, clip.duration, 37, dtype=np.float32)
    poses = context.decompress_tracks(np.full(times.size, handle, dtype=np.uint32), times)
    exact = True
    tiny_w = 0
    for i, t in enumerate(times):
        expected = ob.oracle_decompress_tracks(aligned, float(t))
        exact = exact and np.array_equal(poses[i].view(np.uint32)[:, [0, 1, 2, 3, 4, 5, 6, 8, 9, 10]], expected.view(np.uint32)[:, [0, 1, 2, 3, 4, 5, 6, 8, 9, 10]])
        tiny_w += int(np.any((np.abs(expected[:, 3]) > 0) & (np.abs(expected[:, 3]) < 1e-13)))
    print("TINY_W", tiny_w, "EXACT", int(exact), "SHORT", runtime.analyze_clip(aligned, check_hash=False) & runtime.CLIP_FACT_SHORT_EXACT_MATH)
"""


def _run(extra_env):
    env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""), **extra_env)
    completed = subprocess.run([sys.executable, "-c", _SCRIPT], cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert completed.returncode == 0, completed.stderr[-2000:]
    line = next(l for l in completed.stdout.splitlines() if l.startswith("TINY_W"))
    fields = line.split()
    return int(fields[1]), int(fields[3]), int(fields[5])


def test_a_clip_that_reaches_the_gap_of_the_short_forms_is_recognised():
    tiny_w, exact, short = _run({})
    assert tiny_w > 0                   # the clip does what it was built for: W = sqrt(1e-40)
    assert short == 0                   # ... registration sees it ...
    assert exact == 1                   # ... and its poses are the oracle's bit for bit (the compiler's square root ran)


def test_the_analysis_is_what_keeps_the_bits():
    lab_library = os.path.join(ROOT, "acl_amd", "lib", "libaclhip_lab.so")
    assert os.path.exists(lab_library), "acl_amd/build.py builds it (build_hip_lab)"
    tiny_w, exact, short = _run({"ACLHIP_SHORT_EXACT_MATH": "1", "ACLHIP_LIBRARY": lab_library})
    assert tiny_w > 0 and exact == 0    # overruled: the short form meets an argument below 2^-96 and rounds it differently
    assert short == 0                   # aclhip_analyze_clip still reports the analysis, not the override


def test_the_shipped_library_does_not_listen_to_the_override():
    tiny_w, exact, short = _run({"ACLHIP_SHORT_EXACT_MATH": "1"})
    assert tiny_w > 0 and exact == 1 and short == 0


def test_the_smallest_components_the_analysis_lets_through_are_safe():
    """x = 1 exactly next to y = 1e-14 (just above the 2^-47 the analysis asks of a nonzero component): W^2 = 1e-28 behind an exact
    cancellation, still >= 2^-96 -- the clip keeps the short forms and its poses the oracle's bits"""
    tiny_w, exact, short = _run({"ACLHIP_TEST_TINY_COMPONENT": "1.0e-14"})
    assert tiny_w > 0 and short != 0 and exact == 1


@pytest.mark.parametrize("normalization", [0, 1])
def test_raw_rotations_reach_the_walk_bit_exact(normalization):
    """Half of the rotation sub-tracks stored raw: their waves decode with the compiler's forms, the others with the short ones, and
    the object space walk takes the short normalize only when the decode normalized what it hands over (k_clip_raw_rotations,
    kernels_consumers.inl: walk_may_use_short_exact_math). Bit identical to the oracle either way."""
    from acl_amd import runtime, synth
    from oracle import bindings as ob
    import helpers
    rng = np.random.default_rng(91 + normalization)
    clips = [synth.build_clip(seed=610 + k, num_tracks=140, num_samples=30 + k, raw_fraction=0.5, has_scale=1, scale_default=0.6) for k in range(2)]
    blobs = [c.blob for c in clips]
    parents = np.zeros(140, dtype=np.uint32)
    parents[0] = runtime.NO_PARENT
    for i in range(1, 140):
        parents[i] = rng.integers(max(0, i - 6), i)
    with runtime.Context(0) as context:
        handles = np.array([context.register_clip(b) for b in blobs], dtype=np.uint32)
        for handle in handles:
            context.set_clip_hierarchy(int(handle), parents)
        which = rng.integers(0, 2, size=48)
        times = np.array([rng.uniform(0.0, clips[c].duration) for c in which], dtype=np.float32)
        params = runtime.default_params(normalization=normalization)
        options = ob.default_options(normalization=normalization)
        got = context.decompress_poses(handles[which], times, object_space=True, params=params)
        plain = context.decompress_tracks(handles[which], times, params=params)
        for i in range(which.size):
            local = ob.oracle_decompress_tracks(blobs[which[i]], float(times[i]), 0, options)
            assert helpers.exact(plain[i], local), i
            assert helpers.exact(got[i], ob.oracle_local_to_object_space(parents, local)), i
        assert context.rejected_instance_count() == 0


def test_a_grid_next_to_zero_whose_key_frames_stay_out_of_the_gap_keeps_the_short_forms():
    """y = 1e-20 again, but next to x = 1 - 2^-24: no exact cancellation in front of it, W^2 = 1.2e-7. The grid test of registration
    refuses (a value within 2^-47 of zero), its exact test on the stored key frames accepts -- and the poses are the oracle's bits"""
    tiny_w, exact, short = _run({"ACLHIP_TEST_X": repr(float(np.nextafter(np.float32(1.0), np.float32(0.0))))})
    assert short != 0 and exact == 1

