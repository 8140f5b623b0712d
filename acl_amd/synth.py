"""ctypes binding of the synthetic clip writer (acl_amd/csrc/clip_synth.cpp -> acl_amd/lib/libaclsynth.so).

Produces legal ACL ``compressed_tracks`` blobs (the layout of
/root/reference/includes/acl/core/impl/compressed_headers.h) for benchmarks and tests. Host only.
"""
import ctypes
import os

import numpy as np

_LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libaclsynth.so")
_lib = None


class ClipSpec(ctypes.Structure):
    """Mirror of ``aclsynth_spec`` (acl_amd/csrc/clip_synth.h)."""
    _fields_ = [
        ("seed", ctypes.c_uint32), ("num_tracks", ctypes.c_uint32), ("num_samples", ctypes.c_uint32), ("sample_rate", ctypes.c_float),
        ("version", ctypes.c_uint32), ("has_scale", ctypes.c_uint32), ("default_scale", ctypes.c_uint32), ("wrap", ctypes.c_uint32),
        ("strip_keyframes", ctypes.c_uint32), ("strip_fraction", ctypes.c_float),
        ("rotation_default", ctypes.c_float), ("rotation_constant", ctypes.c_float),
        ("translation_default", ctypes.c_float), ("translation_constant", ctypes.c_float),
        ("scale_default", ctypes.c_float), ("scale_constant", ctypes.c_float),
        ("min_bits", ctypes.c_uint32), ("max_bits", ctypes.c_uint32), ("width0_fraction", ctypes.c_float), ("raw_fraction", ctypes.c_float),
        ("translation_extent", ctypes.c_float), ("ideal_segment_samples", ctypes.c_uint32), ("max_segment_samples", ctypes.c_uint32),
    ]


def _load():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            raise RuntimeError(f"{_LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` first")
        lib = ctypes.CDLL(_LIB_PATH)
        lib.aclsynth_default_spec.argtypes = [ctypes.POINTER(ClipSpec)]
        lib.aclsynth_default_spec.restype = None
        lib.aclsynth_build_clip.argtypes = [ctypes.POINTER(ClipSpec), ctypes.c_void_p, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
        lib.aclsynth_build_clip.restype = ctypes.c_uint32
        _lib = lib
    return _lib


def aligned_bytes(size, alignment=16):
    """A zeroed uint8 numpy array of ``size`` bytes whose data pointer is ``alignment`` aligned (blobs must be 16 byte aligned)."""
    raw = np.zeros(size + alignment, dtype=np.uint8)
    offset = (-raw.ctypes.data) % alignment
    return raw[offset:offset + size]


def default_spec(**overrides):
    spec = ClipSpec()
    _load().aclsynth_default_spec(ctypes.byref(spec))
    for key, value in overrides.items():
        if not hasattr(spec, key):
            raise AttributeError(f"aclsynth_spec has no field '{key}'")
        setattr(spec, key, value)
    return spec


class SyntheticClip:
    """A generated clip: ``blob`` (aligned uint8 array) plus optional side data for tests."""

    def __init__(self, spec, blob, expected, stored, raw):
        self.spec = spec
        self.blob = blob
        self.expected_keyframes = expected    # [num_samples, num_tracks, 12] value each sub-track decodes to AT a stored keyframe
        self.stored_keyframes = stored        # [num_samples] 1 = keyframe present in the blob
        self.raw_keyframes = raw              # [num_samples, num_tracks, 12] lossless source
        self.num_tracks = spec.num_tracks
        self.num_samples = spec.num_samples
        self.sample_rate = spec.sample_rate

    @property
    def duration(self):
        n = self.num_samples + (1 if self.spec.wrap and self.spec.version > 7 and self.num_samples else 0)
        return 0.0 if n <= 1 else float(np.float32(n - 1) / np.float32(self.sample_rate))


def build_clip(spec=None, with_side_data=False, **overrides):
    """Builds one clip. ``spec`` is a ClipSpec (or None for the CMU-shaped default); keyword overrides patch fields."""
    lib = _load()
    if spec is None:
        spec = default_spec(**overrides)
    elif overrides:
        for key, value in overrides.items():
            setattr(spec, key, value)

    size = lib.aclsynth_build_clip(ctypes.byref(spec), None, 0, None, None, None)
    if size == 0:
        raise ValueError("invalid aclsynth_spec")
    blob = aligned_bytes(size)
    expected = stored = raw = None
    if with_side_data:
        expected = np.zeros((spec.num_samples, spec.num_tracks, 12), dtype=np.float32)
        stored = np.zeros(spec.num_samples, dtype=np.uint8)
        raw = np.zeros((spec.num_samples, spec.num_tracks, 12), dtype=np.float32)
    written = lib.aclsynth_build_clip(
        ctypes.byref(spec), blob.ctypes.data, size,
        expected.ctypes.data if with_side_data else None,
        stored.ctypes.data if with_side_data else None,
        raw.ctypes.data if with_side_data else None)
    assert written == size
    return SyntheticClip(spec, blob, expected, stored, raw)
