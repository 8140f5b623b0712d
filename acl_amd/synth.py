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
        ("mirrored_scale_fraction", ctypes.c_float),
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


# ---- scalar track lists (float1f / float2f / float3f / float4f / vector4f) --------------------------------------------
# Written in numpy following the reference's writer (compression/impl/compress.scalar.impl.h:57-215 for the layout,
# write_track_data_impl.h for the streams): raw_buffer_header | tracks_header | scalar_tracks_header | one bit rate byte per
# track | constant values | range values (min[C], extent[C] per quantized track) | animated values, frame major, MSB first
# | 15 bytes of padding.
TRACK_TYPE_COMPONENTS = {0: 1, 1: 2, 2: 3, 3: 4, 4: 4}
_BIT_RATE_NUM_BITS_V0 = [0, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 32]             # core/impl/variable_bit_rates.h:42
_BIT_RATE_NUM_BITS = [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22, 23, 32]


def _hash32(data):
    acc = 2166136261
    for byte in bytes(data):
        acc = ((acc ^ byte) * 16777619) & 0xFFFFFFFF
    return acc


class SyntheticScalarClip:
    """blob: the compressed_tracks bytes (16 byte aligned); keyframes: [num_samples, num_tracks, C] float32, what a decoder returns at each sample."""

    def __init__(self, blob, keyframes, track_type, sample_rate, bit_rates, wrap):
        self.blob = blob
        self.keyframes = keyframes
        self.track_type = track_type
        self.num_samples, self.num_tracks, self.num_components = keyframes.shape
        self.sample_rate = sample_rate
        self.bit_rates = bit_rates
        self.wrap = wrap
        samples = self.num_samples + (1 if wrap and self.num_samples else 0)
        self.duration = float(np.float32(samples - 1) / np.float32(sample_rate)) if samples > 1 else 0.0


def build_scalar_clip(seed=1, track_type=0, num_tracks=16, num_samples=40, sample_rate=30.0, version=10, wrap=0,
                      constant_fraction=0.2, raw_fraction=0.1, min_bits=1, max_bits=23):
    """A legal scalar compressed_tracks blob with every bit rate class: constant (rate 0), quantized (min_bits..max_bits) and raw."""
    import struct
    rng = np.random.default_rng(seed)
    components = TRACK_TYPE_COMPONENTS[track_type]
    table = _BIT_RATE_NUM_BITS_V0 if version == 7 else _BIT_RATE_NUM_BITS
    quantized_rates = [rate for rate, bits in enumerate(table) if 0 < bits < 32 and min_bits <= bits <= max_bits]
    if num_tracks == 0:
        num_samples = 0

    bit_rates, constants, ranges = [], [], []
    keyframes = np.zeros((num_samples, num_tracks, components), dtype=np.float32)
    codes = []          # per track: None | int array [num_samples, C] | float32 array [num_samples, C] (raw)
    for track in range(num_tracks):
        pick = rng.uniform()
        if pick < constant_fraction or num_samples == 0:
            value = rng.uniform(-10.0, 10.0, size=components).astype(np.float32)
            bit_rates.append(0)
            constants.append(value)
            keyframes[:, track] = value
            codes.append(None)
        elif pick < constant_fraction + raw_fraction:
            values = rng.uniform(-1000.0, 1000.0, size=(num_samples, components)).astype(np.float32)
            bit_rates.append(len(table) - 1)
            keyframes[:, track] = values
            codes.append(values)
        else:
            rate = int(rng.choice(quantized_rates))
            num_bits = table[rate]
            range_min = rng.uniform(-50.0, 50.0, size=components).astype(np.float32)
            range_extent = rng.uniform(0.001, 100.0, size=components).astype(np.float32)
            quantized = rng.integers(0, 1 << num_bits, size=(num_samples, components), dtype=np.int64)
            quantized[0], quantized[-1] = 0, (1 << num_bits) - 1                     # both ends of the range appear
            inv_max = np.float32(1.0) / np.float32((1 << num_bits) - 1)
            normalized = quantized.astype(np.float32) * inv_max                      # unpack_*_uXX: float(int) * (1 / max)
            keyframes[:, track] = normalized * range_extent + range_min              # mul, then add (fp32 each)
            bit_rates.append(rate)
            ranges.append(np.concatenate([range_min, range_extent]))
            codes.append(quantized)

    # animated values: frame major, one sample per track, MSB first
    stream = 0
    stream_bits = 0
    for sample in range(num_samples):
        for track in range(num_tracks):
            num_bits = table[bit_rates[track]]
            if num_bits == 0:
                continue
            for c in range(components):
                if num_bits == 32:
                    value = int(np.frombuffer(np.float32(codes[track][sample, c]).tobytes(), dtype=np.uint32)[0])
                else:
                    value = int(codes[track][sample, c])
                stream = (stream << num_bits) | value
                stream_bits += num_bits
    num_bits_per_frame = stream_bits // num_samples if num_samples else 0
    animated_size = (stream_bits + 7) // 8
    animated = (stream << (animated_size * 8 - stream_bits)).to_bytes(animated_size, "big") if animated_size else b""

    constant_bytes = b"".join(np.asarray(v, dtype="<f4").tobytes() for v in constants)
    range_bytes = b"".join(np.asarray(v, dtype="<f4").tobytes() for v in ranges)
    metadata_offset = 20                                                            # sizeof(scalar_tracks_header)
    constants_offset = (metadata_offset + num_tracks + 3) & ~3
    ranges_offset = constants_offset + len(constant_bytes)
    animated_offset = ranges_offset + len(range_bytes)
    size = 8 + 24 + animated_offset + animated_size + 15

    blob = aligned_bytes(size)
    misc_packed = (1 << 30) if wrap else 0
    struct.pack_into("<IHBBIIfI", blob, 8, 0xAC11AC11, version, 0, track_type, num_tracks, num_samples, sample_rate if num_tracks else 0.0, misc_packed)
    struct.pack_into("<5I", blob, 32, num_bits_per_frame, metadata_offset, constants_offset, ranges_offset, animated_offset)
    blob[32 + metadata_offset: 32 + metadata_offset + num_tracks] = np.array(bit_rates, dtype=np.uint8)
    blob[32 + constants_offset: 32 + constants_offset + len(constant_bytes)] = np.frombuffer(constant_bytes, dtype=np.uint8)
    blob[32 + ranges_offset: 32 + ranges_offset + len(range_bytes)] = np.frombuffer(range_bytes, dtype=np.uint8)
    blob[32 + animated_offset: 32 + animated_offset + animated_size] = np.frombuffer(animated, dtype=np.uint8)
    struct.pack_into("<II", blob, 0, size, _hash32(blob[8:size]))
    # the wrap flag is only honoured from v02_01_99 on (compressed_tracks::get_looping_policy, core/impl/compressed_tracks.impl.h:102-110)
    return SyntheticScalarClip(blob, keyframes, track_type, sample_rate, bit_rates, bool(wrap) and version > 7)


def humanoid_hierarchy(num_tracks=100):
    """Parent index per transform of a game-character-like skeleton, sorted parent first (NO_PARENT = 0xFFFFFFFF for the root):
    root > pelvis > 4 spine bones > neck > head with 12 face bones; two arms (clavicle > upper arm > lower arm > hand, 5 fingers of
    3 joints, 2 twist bones); two legs (thigh > calf > foot > ball, 2 twist bones); the rest are 2-bone accessory chains hanging off
    the trunk. 100 transforms: 13 depths, 4 to 12 transforms wide. Smaller / larger counts truncate / add accessory chains."""
    parents = [0xFFFFFFFF, 0]                       # root, pelvis
    spine = []
    for _ in range(4):
        parents.append(spine[-1] if spine else 1)
        spine.append(len(parents) - 1)
    parents.append(spine[-1]); neck = len(parents) - 1
    parents.append(neck); head = len(parents) - 1
    parents += [head] * 12
    for _ in range(2):                              # arms
        parents.append(spine[-1]); clavicle = len(parents) - 1
        parents.append(clavicle); upper = len(parents) - 1
        parents.append(upper); lower = len(parents) - 1
        parents.append(lower); hand = len(parents) - 1
        parents += [upper, lower]                   # twist bones
        for _ in range(5):
            parents.append(hand)
            parents.append(len(parents) - 1)
            parents.append(len(parents) - 1)
    for _ in range(2):                              # legs
        parents.append(1); thigh = len(parents) - 1
        parents.append(thigh); calf = len(parents) - 1
        parents.append(calf); foot = len(parents) - 1
        parents.append(foot)
        parents += [thigh, calf]
    anchors = [1] + spine + [head]
    k = 0
    while len(parents) < num_tracks:
        parents.append(anchors[k % len(anchors)])
        if len(parents) < num_tracks:
            parents.append(len(parents) - 1)
        k += 1
    return np.array(parents[:num_tracks], dtype=np.uint32)
