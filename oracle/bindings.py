"""TEST INFRASTRUCTURE -- ctypes bindings of the CPU oracle (oracle/libacloracle.so) and, when present, of the
reference's own decoder built from /root/reference (oracle/_ref/libaclref.so).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the product
(acl_amd/) never does.
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_PATH = os.path.join(_HERE, "libacloracle.so")
REF_PATH = os.path.join(_HERE, "_ref", "libaclref.so")
REF_ASSERT_PATH = os.path.join(_HERE, "_ref", "libaclref_assert.so")
REF_COMPRESS_PATH = os.path.join(_HERE, "_ref", "libaclref_compress.so")

ROUND_NONE, ROUND_FLOOR, ROUND_CEIL, ROUND_NEAREST, ROUND_PER_TRACK = 0, 1, 2, 3, 4
LOOP_CLAMP, LOOP_WRAP, LOOP_AS_COMPRESSED = 0, 1, 2
NORMALIZE_NEVER, NORMALIZE_LERP_ONLY, NORMALIZE_ALWAYS = 0, 1, 2
DEFAULT_SKIPPED, DEFAULT_CONSTANT, DEFAULT_VARIABLE, DEFAULT_LEGACY = 0, 1, 2, 3


class Database(ctypes.Structure):
    _fields_ = [("clip_segment_headers", ctypes.c_void_p), ("bulk_data", ctypes.c_void_p * 2)]


class Options(ctypes.Structure):
    _fields_ = [
        ("looping_policy", ctypes.c_uint8), ("normalization", ctypes.c_uint8), ("per_track_rounding", ctypes.c_uint8),
        ("default_rotation_mode", ctypes.c_uint8), ("default_translation_mode", ctypes.c_uint8), ("default_scale_mode", ctypes.c_uint8),
        ("default_values", ctypes.c_void_p), ("track_rounding", ctypes.c_void_p), ("database", ctypes.POINTER(Database)),
    ]


class SeekResult(ctypes.Structure):
    _fields_ = [
        ("sample_time", ctypes.c_float), ("interpolation_alpha", ctypes.c_float),
        ("key_frames", ctypes.c_uint32 * 2), ("segment_indices", ctypes.c_uint32 * 2), ("segment_key_frames", ctypes.c_uint32 * 2),
        ("key_frame_bit_offsets", ctypes.c_uint32 * 2),
        ("format_per_track_data", ctypes.c_void_p * 2), ("segment_range_data", ctypes.c_void_p * 2), ("animated_track_data", ctypes.c_void_p * 2),
        ("animated_rotation_bit_size", ctypes.c_uint32 * 2), ("animated_translation_bit_size", ctypes.c_uint32 * 2),
        ("uses_single_segment", ctypes.c_uint8),
    ]


_oracle = None
_ref = {}


def oracle():
    global _oracle
    if _oracle is None:
        if not os.path.exists(ORACLE_PATH):
            raise RuntimeError(f"{ORACLE_PATH} is missing: run `make -C oracle` (or __graft_entry__.build())")
        lib = ctypes.CDLL(ORACLE_PATH)
        vp, f32, i32, u32, u64 = ctypes.c_void_p, ctypes.c_float, ctypes.c_int, ctypes.c_uint32, ctypes.c_uint64
        lib.aclo_default_options.argtypes = [ctypes.POINTER(Options)]
        lib.aclo_is_valid.argtypes = [vp, u64, i32]
        lib.aclo_hash32.argtypes = [vp, u64]
        lib.aclo_hash32.restype = u32
        lib.aclo_num_tracks.argtypes = [vp]
        lib.aclo_num_tracks.restype = u32
        lib.aclo_num_samples.argtypes = [vp]
        lib.aclo_num_samples.restype = u32
        lib.aclo_sample_rate.argtypes = [vp]
        lib.aclo_sample_rate.restype = f32
        lib.aclo_finite_duration.argtypes = [vp, i32]
        lib.aclo_finite_duration.restype = f32
        pu32, pf32 = ctypes.POINTER(u32), ctypes.POINTER(f32)
        lib.aclo_find_linear_interpolation_samples_with_sample_rate.argtypes = [u32, f32, f32, i32, i32, pu32, pu32, pf32]
        lib.aclo_find_linear_interpolation_samples_with_duration.argtypes = [u32, f32, f32, i32, i32, pu32, pu32, pf32]
        lib.aclo_find_linear_interpolation_alpha.argtypes = [f32, u32, u32, i32, i32]
        lib.aclo_find_linear_interpolation_alpha.restype = f32
        lib.aclo_apply_rounding_policy.argtypes = [f32, i32]
        lib.aclo_apply_rounding_policy.restype = f32
        lib.aclo_unpack_vector3_uXX.argtypes = [u32, vp, u32, vp]
        lib.aclo_unpack_vector3_96.argtypes = [vp, u32, vp]
        lib.aclo_unpack_vector3_u48.argtypes = [vp, vp]
        lib.aclo_unpack_vector3_u24.argtypes = [vp, vp]
        lib.aclo_pack_vector3_uXX.argtypes = [vp, u32, vp]
        lib.aclo_memcpy_bits.argtypes = [vp, u64, vp, u64, u64]
        lib.aclo_selftest_pack_vector3_uXX.argtypes = [u32, u32]
        lib.aclo_selftest_pack_vector3_uXX.restype = u32
        lib.aclo_seek.argtypes = [vp, f32, i32, ctypes.POINTER(Options), ctypes.POINTER(SeekResult)]
        lib.aclo_decompress_tracks.argtypes = [vp, f32, i32, ctypes.POINTER(Options), vp]
        lib.aclo_decompress_track.argtypes = [vp, f32, i32, ctypes.POINTER(Options), u32, vp]
        lib.aclo_decompress_tracks_batch.argtypes = [vp, vp, vp, u32, i32, ctypes.POINTER(Options), vp, u64]
        lib.aclo_scalar_num_components.argtypes = [vp]
        lib.aclo_scalar_num_components.restype = u32
        lib.aclo_scalar_decompress_tracks.argtypes = [vp, f32, i32, ctypes.POINTER(Options), vp]
        lib.aclo_scalar_decompress_track.argtypes = [vp, f32, i32, ctypes.POINTER(Options), u32, vp]
        lib.aclo_calculate_num_samples.argtypes = [f32, f32]
        lib.aclo_calculate_num_samples.restype = u32
        lib.aclo_calculate_duration.argtypes = [u32, f32]
        lib.aclo_calculate_duration.restype = f32
        lib.aclo_calculate_finite_duration.argtypes = [u32, f32]
        lib.aclo_calculate_finite_duration.restype = f32
        for name in ("aclo_count_set_bits", "aclo_count_leading_zeros", "aclo_count_trailing_zeros"):
            getattr(lib, name).argtypes = [u32]
            getattr(lib, name).restype = u32
        lib.aclo_pack_scalar_unsigned.argtypes = [f32, u32]
        lib.aclo_pack_scalar_unsigned.restype = u32
        lib.aclo_pack_scalar_signed.argtypes = [f32, u32]
        lib.aclo_pack_scalar_signed.restype = u32
        lib.aclo_unpack_scalar_unsigned.argtypes = [u32, u32]
        lib.aclo_unpack_scalar_unsigned.restype = f32
        lib.aclo_unpack_scalar_signed.argtypes = [u32, u32]
        lib.aclo_unpack_scalar_signed.restype = f32
        lib.aclo_unpack_scalarf_32.argtypes = [vp, u32]
        lib.aclo_unpack_scalarf_32.restype = f32
        lib.aclo_unpack_scalarf_uXX.argtypes = [u32, vp, u32]
        lib.aclo_unpack_scalarf_uXX.restype = f32
        lib.aclo_selftest_scalar_packing.argtypes = [u32, u32]
        lib.aclo_selftest_scalar_packing.restype = u32
        lib.aclo_quat_mul.argtypes = [vp, vp, vp]
        lib.aclo_quat_mul.restype = None
        lib.aclo_qvv_mul.argtypes = [vp, vp, vp]
        lib.aclo_qvv_mul.restype = None
        lib.aclo_apply_additive_to_base.argtypes = [i32, vp, vp, u32, vp]
        lib.aclo_apply_additive_to_base.restype = None
        lib.aclo_local_to_object_space.argtypes = [vp, vp, u32, vp]
        lib.aclo_local_to_object_space.restype = None
        _oracle = lib
    return _oracle


def have_ref(asserting=False):
    return os.path.exists(REF_ASSERT_PATH if asserting else REF_PATH)


def ref(asserting=False):
    """The reference's own decoder (oracle/_ref/libaclref*.so). Raises if it was never built."""
    key = bool(asserting)
    if key not in _ref:
        path = REF_ASSERT_PATH if asserting else REF_PATH
        if not os.path.exists(path):
            raise RuntimeError(f"{path} is missing: it is built from /root/reference by `make -C oracle ref`")
        lib = ctypes.CDLL(path)
        vp, f32, i32, u32 = ctypes.c_void_p, ctypes.c_float, ctypes.c_int, ctypes.c_uint32
        lib.aclref_is_valid.argtypes = [vp, i32]
        lib.aclref_get_duration.argtypes = [vp, i32]
        lib.aclref_get_duration.restype = f32
        lib.aclref_get_num_tracks.argtypes = [vp]
        lib.aclref_get_num_tracks.restype = u32
        lib.aclref_get_num_samples.argtypes = [vp]
        lib.aclref_get_num_samples.restype = u32
        lib.aclref_decompress.argtypes = [vp, f32, i32, i32, i32, i32, i32, vp, vp, vp]
        lib.aclref_decompress_many.argtypes = [vp, vp, u32, i32, i32, i32, i32, vp]
        lib.aclref_bench.argtypes = [vp, vp, vp, u32, u32, u32, u32, vp]
        lib.aclref_bench.restype = ctypes.c_double
        lib.aclref_bench_timed.argtypes = [vp, vp, vp, u32, u32, u32, ctypes.c_double, i32, vp]
        lib.aclref_bench_timed.restype = ctypes.c_double
        if hasattr(lib, "aclref_get_metadata"):
            lib.aclref_get_metadata.argtypes = [vp, vp, vp, vp, vp]
        if hasattr(lib, "aclref_bench_cold"):
            lib.aclref_bench_cold.argtypes = [vp, u32, vp, u32, u32, u32, ctypes.c_uint64, ctypes.c_double]
            lib.aclref_bench_cold.restype = ctypes.c_double
        _ref[key] = lib
    return _ref[key]


def default_options(**overrides):
    options = Options()
    oracle().aclo_default_options(ctypes.byref(options))
    for key, value in overrides.items():
        setattr(options, key, value)
    return options


def _ptr(array):
    return array.ctypes.data if array is not None else None


def oracle_decompress_tracks(blob, sample_time, rounding=ROUND_NONE, options=None, out=None):
    """seek + decompress_tracks through the C restatement. Returns [num_tracks, 12] float32."""
    lib = oracle()
    num_tracks = lib.aclo_num_tracks(blob.ctypes.data)
    if out is None:
        out = np.zeros((num_tracks, 12), dtype=np.float32)
    if options is None:
        options = default_options()
    result = lib.aclo_decompress_tracks(blob.ctypes.data, ctypes.c_float(sample_time), rounding, ctypes.byref(options), out.ctypes.data)
    if result != 0:
        raise RuntimeError(f"aclo_decompress_tracks failed: {result}")
    return out


def oracle_decompress_tracks_batch(blobs, clip_indices, sample_times, max_tracks, rounding=ROUND_NONE, options=None, out=None, threads=None):
    """seek + decompress_tracks for EVERY instance (blobs[clip_indices[i]], sample_times[i]) through the C restatement
    (aclo_decompress_tracks_batch), split over host threads (ctypes releases the GIL). Returns [count, max_tracks, 12] float32;
    rows of clips with fewer tracks keep what `out` held (zeros when it is allocated here)."""
    import concurrent.futures
    import os
    lib = oracle()
    count = int(len(clip_indices))
    indices = np.ascontiguousarray(clip_indices, dtype=np.uint32)
    times = np.ascontiguousarray(sample_times, dtype=np.float32)
    if out is None:
        out = np.zeros((count, max_tracks, 12), dtype=np.float32)
    if options is None:
        options = default_options()
    blob_ptrs = (ctypes.c_void_p * len(blobs))(*[b.ctypes.data for b in blobs])
    if threads is None:
        threads = max(1, min(64, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)))
    piece = max(256, (count + threads - 1) // threads)

    def run(first):
        n = min(piece, count - first)
        result = lib.aclo_decompress_tracks_batch(blob_ptrs, indices[first:].ctypes.data, times[first:].ctypes.data, n, rounding, ctypes.byref(options),
                                                  out[first:].ctypes.data, max_tracks * 12)
        if result != 0:
            raise RuntimeError(f"aclo_decompress_tracks_batch failed: {result}")

    with concurrent.futures.ThreadPoolExecutor(max_workers=threads) as pool:
        list(pool.map(run, range(0, count, piece)))
    return out


def _host_threads(threads):
    import os
    if threads is not None:
        return threads
    return max(1, min(64, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)))


def oracle_decompress_poses_batch(blobs, clip_indices, sample_times, num_transforms, additive_format=0, base_clip_indices=None, base_sample_times=None,
                                  parent_indices=None, rounding=ROUND_NONE, options=None, threads=None):
    """decode -> apply_additive_to_base -> local_to_object_space for EVERY instance (aclo_decompress_poses_batch), split over host
    threads. Returns [count, num_transforms, 12] float32."""
    import concurrent.futures
    lib = oracle()
    lib.aclo_decompress_poses_batch.restype = ctypes.c_int
    count = int(len(clip_indices))
    indices = np.ascontiguousarray(clip_indices, dtype=np.uint32)
    times = np.ascontiguousarray(sample_times, dtype=np.float32)
    base_indices = np.ascontiguousarray(base_clip_indices if base_clip_indices is not None else np.zeros(count), dtype=np.uint32)
    base_times = np.ascontiguousarray(base_sample_times if base_sample_times is not None else np.zeros(count), dtype=np.float32)
    parents = None if parent_indices is None else np.ascontiguousarray(parent_indices, dtype=np.uint32)
    out = np.zeros((count, num_transforms, 12), dtype=np.float32)
    if options is None:
        options = default_options()
    blob_ptrs = (ctypes.c_void_p * len(blobs))(*[b.ctypes.data for b in blobs])
    threads = _host_threads(threads)
    piece = max(64, (count + threads - 1) // threads)

    def run(first):
        n = min(piece, count - first)
        result = lib.aclo_decompress_poses_batch(blob_ptrs, ctypes.c_void_p(indices[first:].ctypes.data), ctypes.c_void_p(times[first:].ctypes.data), ctypes.c_uint32(n),
                                                 ctypes.c_int(rounding), ctypes.byref(options), ctypes.c_int(int(additive_format)),
                                                 ctypes.c_void_p(base_indices[first:].ctypes.data), ctypes.c_void_p(base_times[first:].ctypes.data),
                                                 ctypes.c_void_p(parents.ctypes.data if parents is not None else None), ctypes.c_uint32(num_transforms),
                                                 ctypes.c_void_p(out[first:].ctypes.data), ctypes.c_uint64(num_transforms * 12))
        if result != 0:
            raise RuntimeError(f"aclo_decompress_poses_batch failed: {result}")

    with concurrent.futures.ThreadPoolExecutor(max_workers=threads) as pool:
        list(pool.map(run, range(0, count, piece)))
    return out


def oracle_blend_poses(poses, weights):
    """aclo_blend_poses: poses [K, num_transforms, 12], weights [K] -> [num_transforms, 12]."""
    lib = oracle()
    poses = [np.ascontiguousarray(p, dtype=np.float32) for p in poses]
    weights = np.ascontiguousarray(weights, dtype=np.float32)
    num_transforms = poses[0].shape[0]
    out = np.zeros((num_transforms, 12), dtype=np.float32)
    pointers = (ctypes.c_void_p * len(poses))(*[p.ctypes.data for p in poses])
    lib.aclo_blend_poses.restype = None
    lib.aclo_blend_poses(pointers, ctypes.c_void_p(weights.ctypes.data), ctypes.c_uint32(len(poses)), ctypes.c_uint32(num_transforms), ctypes.c_void_p(out.ctypes.data))
    return out


def oracle_decompress_blended_poses_batch(blobs, clip_indices, sample_times, blend_clip_indices, blend_sample_times, blend_weights, num_transforms,
                                          additive_format=0, base_clip_indices=None, base_sample_times=None, parent_indices=None, rounding=ROUND_NONE,
                                          options=None, threads=None):
    """decode K clips -> aclo_blend_poses -> apply_additive_to_base -> local_to_object_space for EVERY instance, split over host threads.
    blend_clip_indices / blend_sample_times: [count, K - 1]; blend_weights: [count, K]. Returns [count, num_transforms, 12] float32."""
    import concurrent.futures
    lib = oracle()
    lib.aclo_decompress_blended_poses_batch.restype = ctypes.c_int
    count = int(len(clip_indices))
    indices = np.ascontiguousarray(clip_indices, dtype=np.uint32)
    times = np.ascontiguousarray(sample_times, dtype=np.float32)
    weights = np.ascontiguousarray(blend_weights, dtype=np.float32).reshape(count, -1)
    num_blend = weights.shape[1]
    others = np.ascontiguousarray(blend_clip_indices, dtype=np.uint32).reshape(count, num_blend - 1)
    other_times = np.ascontiguousarray(blend_sample_times, dtype=np.float32).reshape(count, num_blend - 1)
    base_indices = np.ascontiguousarray(base_clip_indices if base_clip_indices is not None else np.zeros(count), dtype=np.uint32)
    base_times = np.ascontiguousarray(base_sample_times if base_sample_times is not None else np.zeros(count), dtype=np.float32)
    parents = None if parent_indices is None else np.ascontiguousarray(parent_indices, dtype=np.uint32)
    out = np.zeros((count, num_transforms, 12), dtype=np.float32)
    if options is None:
        options = default_options()
    blob_ptrs = (ctypes.c_void_p * len(blobs))(*[b.ctypes.data for b in blobs])
    threads = _host_threads(threads)
    piece = max(64, (count + threads - 1) // threads)

    def run(first):
        n = min(piece, count - first)
        result = lib.aclo_decompress_blended_poses_batch(
            blob_ptrs, ctypes.c_void_p(indices[first:].ctypes.data), ctypes.c_void_p(times[first:].ctypes.data), ctypes.c_uint32(n),
            ctypes.c_int(rounding), ctypes.byref(options), ctypes.c_uint32(num_blend), ctypes.c_void_p(others[first:].ctypes.data),
            ctypes.c_void_p(other_times[first:].ctypes.data), ctypes.c_void_p(weights[first:].ctypes.data), ctypes.c_int(int(additive_format)),
            ctypes.c_void_p(base_indices[first:].ctypes.data), ctypes.c_void_p(base_times[first:].ctypes.data),
            ctypes.c_void_p(parents.ctypes.data if parents is not None else None), ctypes.c_uint32(num_transforms),
            ctypes.c_void_p(out[first:].ctypes.data), ctypes.c_uint64(num_transforms * 12))
        if result != 0:
            raise RuntimeError(f"aclo_decompress_blended_poses_batch failed: {result}")

    with concurrent.futures.ThreadPoolExecutor(max_workers=threads) as pool:
        list(pool.map(run, range(0, count, piece)))
    return out


def oracle_scalar_decompress_tracks_batch(blobs, clip_indices, sample_times, row_floats, rounding=ROUND_NONE, options=None, threads=None):
    """scalar decompress_tracks for EVERY instance (aclo_scalar_decompress_tracks_batch), split over host threads. Returns
    [count, row_floats] float32 (num_tracks * C values per instance, the rest of a row zero)."""
    import concurrent.futures
    lib = oracle()
    lib.aclo_scalar_decompress_tracks_batch.restype = ctypes.c_int
    count = int(len(clip_indices))
    indices = np.ascontiguousarray(clip_indices, dtype=np.uint32)
    times = np.ascontiguousarray(sample_times, dtype=np.float32)
    out = np.zeros((count, row_floats), dtype=np.float32)
    if options is None:
        options = default_options()
    blob_ptrs = (ctypes.c_void_p * len(blobs))(*[b.ctypes.data for b in blobs])
    threads = _host_threads(threads)
    piece = max(64, (count + threads - 1) // threads)

    def run(first):
        n = min(piece, count - first)
        result = lib.aclo_scalar_decompress_tracks_batch(blob_ptrs, ctypes.c_void_p(indices[first:].ctypes.data), ctypes.c_void_p(times[first:].ctypes.data), ctypes.c_uint32(n),
                                                         ctypes.c_int(rounding), ctypes.byref(options), ctypes.c_void_p(out[first:].ctypes.data), ctypes.c_uint64(row_floats))
        if result != 0:
            raise RuntimeError(f"aclo_scalar_decompress_tracks_batch failed: {result}")

    with concurrent.futures.ThreadPoolExecutor(max_workers=threads) as pool:
        list(pool.map(run, range(0, count, piece)))
    return out


def oracle_decompress_track(blob, sample_time, track_index, rounding=ROUND_NONE, options=None):
    lib = oracle()
    out = np.zeros(12, dtype=np.float32)
    if options is None:
        options = default_options()
    result = lib.aclo_decompress_track(blob.ctypes.data, ctypes.c_float(sample_time), rounding, ctypes.byref(options), track_index, out.ctypes.data)
    if result != 0:
        raise RuntimeError(f"aclo_decompress_track failed: {result}")
    return out


def ref_decompress(blob, sample_time, rounding=ROUND_NONE, looping=-1, settings=0, default_mode=0, track_index=-1,
                   defaults=None, track_rounding=None, out=None, asserting=False):
    """seek + decompress_tracks (track_index < 0) or decompress_track through the reference's own headers."""
    lib = ref(asserting)
    num_tracks = lib.aclref_get_num_tracks(blob.ctypes.data)
    if out is None:
        out = np.zeros((num_tracks, 12), dtype=np.float32)
    result = lib.aclref_decompress(blob.ctypes.data, ctypes.c_float(sample_time), rounding, looping, settings, default_mode, track_index,
                                   out.ctypes.data, _ptr(defaults), _ptr(track_rounding))
    if result != 0:
        raise RuntimeError(f"aclref_decompress failed: {result}")
    return out


def ref_get_metadata(blob):
    """compressed_tracks::get_parent_track_index / get_track_description of every track through the reference's own accessors:
    (parents, (default_values [n, 12], precisions, shell_distances) or None when the blob stores no descriptions)"""
    lib = ref()
    num_tracks = lib.aclref_get_num_tracks(blob.ctypes.data)
    parents = np.zeros(max(num_tracks, 1), dtype=np.uint32)
    defaults, precisions, shells = np.zeros((max(num_tracks, 1), 12), dtype=np.float32), np.zeros(max(num_tracks, 1), dtype=np.float32), np.zeros(max(num_tracks, 1), dtype=np.float32)
    result = lib.aclref_get_metadata(blob.ctypes.data, parents.ctypes.data, defaults.ctypes.data, precisions.ctypes.data, shells.ctypes.data)
    if result < 0:
        raise RuntimeError("the description's parent index differs from get_parent_track_index")
    return parents[:num_tracks], ((defaults[:num_tracks], precisions[:num_tracks], shells[:num_tracks]) if result == 1 else None)


_ref_compress = None


def have_ref_compressor():
    return os.path.exists(REF_COMPRESS_PATH)


def ref_compress(raw, sample_rate, parents=None, precision=0.0001, shell_distance=1.0, optimize_loops=False, strip_proportion=None, aligned_bytes=None):
    """Compresses raw [num_samples, num_tracks, 12] qvv animation with the REFERENCE's compressor (default settings).
    Returns a 16 byte aligned uint8 array holding the compressed_tracks."""
    global _ref_compress
    if _ref_compress is None:
        if not os.path.exists(REF_COMPRESS_PATH):
            raise RuntimeError(f"{REF_COMPRESS_PATH} is missing: built from /root/reference by `make -C oracle ref`")
        lib = ctypes.CDLL(REF_COMPRESS_PATH)
        lib.aclref_compress.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_float, ctypes.c_void_p, ctypes.c_float, ctypes.c_float,
                                        ctypes.c_uint32, ctypes.c_float, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_char_p, ctypes.c_uint32]
        lib.aclref_compress.restype = ctypes.c_uint32
        _ref_compress = lib
    raw = np.ascontiguousarray(raw, dtype=np.float32)
    num_samples, num_tracks = raw.shape[0], raw.shape[1]
    if parents is None:
        parents = np.arange(-1, num_tracks - 1, dtype=np.int32)    # a chain
    parents = np.ascontiguousarray(parents, dtype=np.int32)
    flags = (1 if optimize_loops else 0) | (2 if strip_proportion is not None else 0)
    error = ctypes.create_string_buffer(256)
    args = [raw.ctypes.data, num_tracks, num_samples, ctypes.c_float(sample_rate), parents.ctypes.data, ctypes.c_float(precision), ctypes.c_float(shell_distance),
            flags, ctypes.c_float(strip_proportion or 0.0)]
    size = _ref_compress.aclref_compress(*args, None, 0, error, 256)
    if size == 0:
        raise RuntimeError(f"aclref_compress failed: {error.value.decode()}")
    if aligned_bytes is None:
        from acl_amd.synth import aligned_bytes as make_aligned
    else:
        make_aligned = aligned_bytes
    blob = make_aligned(size)
    written = _ref_compress.aclref_compress(*args, blob.ctypes.data, size, error, 256)
    assert written == size
    return blob


class RefCompressSettings(ctypes.Structure):
    """aclref_compress_settings (oracle/ref_compress_bridge.cpp)"""
    _fields_ = [("level", ctypes.c_uint32), ("rotation_format", ctypes.c_uint32), ("translation_format", ctypes.c_uint32), ("scale_format", ctypes.c_uint32),
                ("flags", ctypes.c_uint32), ("strip_proportion", ctypes.c_float), ("strip_threshold", ctypes.c_float), ("precision", ctypes.c_float), ("shell_distance", ctypes.c_float)]


LEVELS = {"lowest": 0, "low": 1, "medium": 2, "high": 3, "highest": 4}
ROTATION_FORMATS = {"quatf_full": 0, "quatf_drop_w_full": 2, "quatf_drop_w_variable": 3}
VECTOR_FORMATS = {"vector3f_full": 0, "vector3f_variable": 1}
COMPRESS_FLAGS = {"optimize_loops": 1, "enable_database_support": 2, "include_contributing_error": 4, "include_parent_track_indices": 8, "include_track_descriptions": 16,
                  "include_track_names": 32, "include_track_list_name": 64, "matrix_error_metric": 128, "strip_trivial": 256}
_ref_compress_ex = None


def ref_compress_ex(raw, sample_rate, parents=None, bind_pose=None, level="medium", rotation_format="quatf_drop_w_variable", translation_format="vector3f_variable",
                    scale_format="vector3f_variable", precision=0.0001, shell_distance=1.0, strip_proportion=0.0, strip_threshold=0.0, **flags):
    """compress_track_list with any compression_settings (the reference's regression configs: test_data/configs/*.sjson).
    raw [num_samples, num_tracks, 12]; bind_pose [num_tracks, 12] = track_desc_transformf::default_value; flags: COMPRESS_FLAGS names = True."""
    global _ref_compress_ex
    if _ref_compress_ex is None:
        if not os.path.exists(REF_COMPRESS_PATH):
            raise RuntimeError(f"{REF_COMPRESS_PATH} is missing: built from /root/reference by `make -C oracle ref`")
        lib = ctypes.CDLL(REF_COMPRESS_PATH)
        lib.aclref_compress_ex.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_float, ctypes.c_void_p, ctypes.c_void_p, ctypes.POINTER(RefCompressSettings),
                                           ctypes.c_void_p, ctypes.c_uint32, ctypes.c_char_p, ctypes.c_uint32]
        lib.aclref_compress_ex.restype = ctypes.c_uint32
        _ref_compress_ex = lib
    raw = np.ascontiguousarray(raw, dtype=np.float32)
    num_samples, num_tracks = raw.shape[0], raw.shape[1]
    if parents is None:
        parents = np.arange(-1, num_tracks - 1, dtype=np.int32)    # a chain
    parents = np.ascontiguousarray(parents, dtype=np.int32)
    if bind_pose is not None:
        bind_pose = np.ascontiguousarray(bind_pose, dtype=np.float32)
        assert bind_pose.shape == (num_tracks, 12)
    settings = RefCompressSettings(LEVELS[level], ROTATION_FORMATS[rotation_format], VECTOR_FORMATS[translation_format], VECTOR_FORMATS[scale_format],
                                   sum(COMPRESS_FLAGS[name] for name, on in flags.items() if on), strip_proportion, strip_threshold, precision, shell_distance)
    error = ctypes.create_string_buffer(256)
    args = [raw.ctypes.data, num_tracks, num_samples, ctypes.c_float(sample_rate), parents.ctypes.data, _ptr(bind_pose), ctypes.byref(settings)]
    size = _ref_compress_ex.aclref_compress_ex(*args, None, 0, error, 256)
    if size == 0:
        raise RuntimeError(f"aclref_compress_ex failed: {error.value.decode()}")
    from acl_amd.synth import aligned_bytes
    blob = aligned_bytes(size)
    written = _ref_compress_ex.aclref_compress_ex(*args, blob.ctypes.data, size, error, 256)
    assert written == size
    return blob


REF_DB_PATH = os.path.join(_HERE, "_ref", "libaclref_db.so")
_ref_db = None


# ---- scalar track lists (float1f .. vector4f) ----
REF_SCALAR_PATH = os.path.join(_HERE, "_ref", "libaclref_scalar.so")
TRACK_FLOAT1F, TRACK_FLOAT2F, TRACK_FLOAT3F, TRACK_FLOAT4F, TRACK_VECTOR4F, TRACK_QVVF = 0, 1, 2, 3, 4, 12
_ref_scalar = None


def oracle_scalar_decompress_tracks(blob, sample_time, rounding=ROUND_NONE, options=None, out=None):
    """seek + decompress_tracks of a scalar track list through the C restatement. Returns [num_tracks, num_components] float32."""
    lib = oracle()
    if out is None:
        out = np.zeros((lib.aclo_num_tracks(blob.ctypes.data), lib.aclo_scalar_num_components(blob.ctypes.data)), dtype=np.float32)
    if options is None:
        options = default_options()
    result = lib.aclo_scalar_decompress_tracks(blob.ctypes.data, ctypes.c_float(sample_time), rounding, ctypes.byref(options), out.ctypes.data)
    if result != 0:
        raise RuntimeError(f"aclo_scalar_decompress_tracks failed: {result}")
    return out


def oracle_scalar_decompress_track(blob, sample_time, track_index, rounding=ROUND_NONE, options=None):
    lib = oracle()
    out = np.zeros(lib.aclo_scalar_num_components(blob.ctypes.data), dtype=np.float32)
    if options is None:
        options = default_options()
    result = lib.aclo_scalar_decompress_track(blob.ctypes.data, ctypes.c_float(sample_time), rounding, ctypes.byref(options), track_index, out.ctypes.data)
    if result != 0:
        raise RuntimeError(f"aclo_scalar_decompress_track failed: {result}")
    return out


def have_ref_scalar():
    return os.path.exists(REF_SCALAR_PATH)


def ref_scalar():
    """The reference's scalar track compressor + decoder (oracle/_ref/libaclref_scalar.so)."""
    global _ref_scalar
    if _ref_scalar is None:
        if not os.path.exists(REF_SCALAR_PATH):
            raise RuntimeError(f"{REF_SCALAR_PATH} is missing: built from /root/reference by `make -C oracle ref`")
        lib = ctypes.CDLL(REF_SCALAR_PATH)
        vp, u32, i32, f32 = ctypes.c_void_p, ctypes.c_uint32, ctypes.c_int, ctypes.c_float
        lib.aclref_scalar_decompress.argtypes = [vp, f32, i32, i32, i32, i32, vp, vp]
        lib.aclref_scalar_compress.argtypes = [vp, u32, u32, u32, f32, f32, u32, vp, u32, ctypes.c_char_p, u32]
        lib.aclref_scalar_compress.restype = u32
        lib.aclref_scalar_bench.argtypes = [vp, vp, vp, u32, u32, u32, u32]
        lib.aclref_scalar_bench.restype = ctypes.c_double
        lib.aclref_scalar_bench_timed.argtypes = [vp, vp, vp, u32, u32, u32, ctypes.c_double, i32, vp]
        lib.aclref_scalar_bench_timed.restype = ctypes.c_double
        _ref_scalar = lib
    return _ref_scalar


def ref_scalar_compress(raw, track_type, sample_rate, precision=0.0001, optimize_loops=False):
    """compress_track_list on raw scalar samples [num_samples, num_tracks, num_components]. Returns the 16 byte aligned blob."""
    from acl_amd.synth import aligned_bytes
    lib = ref_scalar()
    raw = np.ascontiguousarray(raw, dtype=np.float32)
    if raw.ndim == 2:
        raw = raw[:, :, None]
    num_samples, num_tracks = raw.shape[0], raw.shape[1]
    error = ctypes.create_string_buffer(256)
    args = (raw.ctypes.data, track_type, num_tracks, num_samples, ctypes.c_float(sample_rate), ctypes.c_float(precision), 1 if optimize_loops else 0)
    size = lib.aclref_scalar_compress(*args, None, 0, error, 256)
    if size == 0:
        raise RuntimeError(f"aclref_scalar_compress failed: {error.value.decode()}")
    blob = aligned_bytes(size)
    assert lib.aclref_scalar_compress(*args, blob.ctypes.data, size, error, 256) == size
    return blob


def ref_scalar_decompress(blob, sample_time, rounding=ROUND_NONE, looping=-1, settings=0, track_index=-1, track_rounding=None, out=None):
    """The reference's decompression_context on a scalar track list. settings 0 = default, 1 = debug (per track rounding)."""
    lib = oracle()
    if out is None:
        out = np.zeros((lib.aclo_num_tracks(blob.ctypes.data), lib.aclo_scalar_num_components(blob.ctypes.data)), dtype=np.float32)
    result = ref_scalar().aclref_scalar_decompress(blob.ctypes.data, ctypes.c_float(sample_time), rounding, looping, settings, track_index, out.ctypes.data, _ptr(track_rounding))
    if result != 0:
        raise RuntimeError(f"aclref_scalar_decompress failed: {result}")
    return out


def have_ref_database():
    return os.path.exists(REF_DB_PATH)


def ref_db():
    """The reference's database builder + database_context (oracle/_ref/libaclref_db.so)."""
    global _ref_db
    if _ref_db is None:
        if not os.path.exists(REF_DB_PATH):
            raise RuntimeError(f"{REF_DB_PATH} is missing: built from /root/reference by `make -C oracle ref`")
        lib = ctypes.CDLL(REF_DB_PATH)
        vp, u32, i32, f32 = ctypes.c_void_p, ctypes.c_uint32, ctypes.c_int, ctypes.c_float
        lib.aclref_db_compress.argtypes = [vp, u32, u32, f32, f32, vp, u32]
        lib.aclref_db_compress.restype = u32
        lib.aclref_db_build.argtypes = [vp, u32, f32, f32, u32]
        lib.aclref_db_build.restype = vp
        for name in ("aclref_db_clip_size",):
            getattr(lib, name).argtypes = [vp, u32]
            getattr(lib, name).restype = u32
        lib.aclref_db_get_clip.argtypes = [vp, u32, vp]
        lib.aclref_db_database_size.argtypes = [vp, i32]
        lib.aclref_db_database_size.restype = u32
        lib.aclref_db_get_database.argtypes = [vp, i32, vp]
        lib.aclref_db_bulk_size.argtypes = [vp, i32]
        lib.aclref_db_bulk_size.restype = u32
        lib.aclref_db_get_bulk.argtypes = [vp, i32, vp]
        lib.aclref_db_num_chunks.argtypes = [vp, i32]
        lib.aclref_db_num_chunks.restype = u32
        lib.aclref_db_context_create.argtypes = [vp]
        lib.aclref_db_stream.argtypes = [vp, i32, u32, i32]
        lib.aclref_db_decompress.argtypes = [vp, u32, f32, i32, vp]
        lib.aclref_db_destroy.argtypes = [vp]
        lib.aclref_db_strip.argtypes = [vp, i32, i32, vp, u32]
        lib.aclref_db_strip.restype = u32
        _ref_db = lib
    return _ref_db


def ref_db_compress(raw, sample_rate, precision=0.0001):
    """compress_track_list with enable_database_support (keeps the contributing error metadata build_database needs)."""
    from acl_amd.synth import aligned_bytes
    lib = ref_db()
    raw = np.ascontiguousarray(raw, dtype=np.float32)
    size = lib.aclref_db_compress(raw.ctypes.data, raw.shape[1], raw.shape[0], ctypes.c_float(sample_rate), ctypes.c_float(precision), None, 0)
    if size == 0:
        raise RuntimeError("aclref_db_compress failed")
    blob = aligned_bytes(size)
    assert lib.aclref_db_compress(raw.ctypes.data, raw.shape[1], raw.shape[0], ctypes.c_float(sample_rate), ctypes.c_float(precision), blob.ctypes.data, size) == size
    return blob


class ReferenceDatabase:
    """build_database + split_database_bulk_data + a database_context with in-memory streamers, all from the reference."""

    def __init__(self, clip_blobs, medium_proportion=0.3, low_proportion=0.4, max_chunk_size=4096):
        from acl_amd.synth import aligned_bytes
        self._lib = ref_db()
        self._inputs = clip_blobs
        pointers = (ctypes.c_void_p * len(clip_blobs))(*[b.ctypes.data for b in clip_blobs])
        self._handle = self._lib.aclref_db_build(pointers, len(clip_blobs), ctypes.c_float(medium_proportion), ctypes.c_float(low_proportion), max_chunk_size)
        if not self._handle:
            raise RuntimeError("build_database failed")
        self.clips = []
        for i in range(len(clip_blobs)):
            blob = aligned_bytes(self._lib.aclref_db_clip_size(self._handle, i))
            self._lib.aclref_db_get_clip(self._handle, i, blob.ctypes.data)
            self.clips.append(blob)
        self.database = aligned_bytes(self._lib.aclref_db_database_size(self._handle, 1))      # bulk data split out
        self._lib.aclref_db_get_database(self._handle, 1, self.database.ctypes.data)
        self.database_inline = aligned_bytes(self._lib.aclref_db_database_size(self._handle, 0))
        self._lib.aclref_db_get_database(self._handle, 0, self.database_inline.ctypes.data)
        self.bulk = {}
        for tier in (1, 2):
            data = aligned_bytes(max(self._lib.aclref_db_bulk_size(self._handle, tier), 1))
            self._lib.aclref_db_get_bulk(self._handle, tier, data.ctypes.data)
            self.bulk[tier] = data[: self._lib.aclref_db_bulk_size(self._handle, tier)]
        self.num_chunks = {tier: self._lib.aclref_db_num_chunks(self._handle, tier) for tier in (1, 2)}
        if self._lib.aclref_db_context_create(self._handle) != 0:
            raise RuntimeError("database_context::initialize failed")

    def stream(self, tier, num_chunks, stream_in=True):
        return self._lib.aclref_db_stream(self._handle, tier, num_chunks, 1 if stream_in else 0)

    def decompress(self, clip_index, sample_time, rounding=ROUND_NONE):
        num_tracks = oracle().aclo_num_tracks(self.clips[clip_index].ctypes.data)
        out = np.zeros((num_tracks, 12), dtype=np.float32)
        result = self._lib.aclref_db_decompress(self._handle, clip_index, ctypes.c_float(sample_time), rounding, out.ctypes.data)
        if result != 0:
            raise RuntimeError(f"aclref_db_decompress failed: {result}")
        return out

    def strip(self, tier, split):
        """strip_database_quality_tier (compression/compress.h:124) of the inline or the split database; None when the reference refuses"""
        from acl_amd.synth import aligned_bytes
        size = self._lib.aclref_db_strip(self._handle, 1 if split else 0, tier, None, 0)
        if size == 0:
            return None
        out = aligned_bytes(size)
        assert self._lib.aclref_db_strip(self._handle, 1 if split else 0, tier, out.ctypes.data, size) == size
        return out

    def close(self):
        if self._handle:
            self._lib.aclref_db_destroy(self._handle)
            self._handle = None


# ---- pose consumers: additive apply and local -> object space ------------------------------------------
REF_POSE_PATH = os.path.join(_HERE, "_ref", "libaclref_pose.so")

ADDITIVE_NONE, ADDITIVE_RELATIVE, ADDITIVE_ADDITIVE0, ADDITIVE_ADDITIVE1 = 0, 1, 2, 3
INVALID_PARENT = 0xFFFFFFFF


def _pose_args(*poses):
    out = []
    for pose in poses:
        pose = np.ascontiguousarray(pose, dtype=np.float32)
        assert pose.ndim == 2 and pose.shape[1] == 12
        out.append(pose)
    return out


def oracle_apply_additive_to_base(additive_format, base_pose, additive_pose):
    """apply_additive_to_base (core/additive_utils.h:150) per transform; poses [num_transforms, 12]"""
    base_pose, additive_pose = _pose_args(base_pose, additive_pose)
    assert base_pose.shape == additive_pose.shape
    out = np.empty_like(base_pose)
    oracle().aclo_apply_additive_to_base(int(additive_format), _ptr(base_pose), _ptr(additive_pose), base_pose.shape[0], _ptr(out))
    return out


def oracle_local_to_object_space(parent_indices, local_pose):
    """local_to_object_space (compression/transform_pose_utils.h:35); pose [num_transforms, 12]"""
    (local_pose,) = _pose_args(local_pose)
    parents = np.ascontiguousarray(parent_indices, dtype=np.uint32)
    assert parents.size == local_pose.shape[0]
    out = np.empty_like(local_pose)
    oracle().aclo_local_to_object_space(_ptr(parents), _ptr(local_pose), local_pose.shape[0], _ptr(out))
    return out


def oracle_quat_mul(lhs, rhs):
    lhs = np.ascontiguousarray(lhs, dtype=np.float32)
    rhs = np.ascontiguousarray(rhs, dtype=np.float32)
    out = np.empty(4, dtype=np.float32)
    oracle().aclo_quat_mul(_ptr(lhs), _ptr(rhs), _ptr(out))
    return out


def have_ref_pose():
    return os.path.exists(REF_POSE_PATH)


def ref_pose():
    """The reference's pose consumers (oracle/_ref/libaclref_pose.so)."""
    if "pose" not in _ref:
        if not have_ref_pose():
            raise RuntimeError(f"{REF_POSE_PATH} missing: run `make -C oracle` where /root/reference exists")
        lib = ctypes.CDLL(REF_POSE_PATH)
        lib.aclref_apply_additive_to_base.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_void_p]
        lib.aclref_apply_additive_to_base.restype = None
        lib.aclref_local_to_object_space.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_void_p]
        lib.aclref_local_to_object_space.restype = None
        _ref["pose"] = lib
    return _ref["pose"]


def ref_apply_additive_to_base(additive_format, base_pose, additive_pose):
    base_pose, additive_pose = _pose_args(base_pose, additive_pose)
    out = np.empty_like(base_pose)
    ref_pose().aclref_apply_additive_to_base(int(additive_format), _ptr(base_pose), _ptr(additive_pose), base_pose.shape[0], _ptr(out))
    return out


def ref_local_to_object_space(parent_indices, local_pose):
    (local_pose,) = _pose_args(local_pose)
    parents = np.ascontiguousarray(parent_indices, dtype=np.uint32)
    out = np.empty_like(local_pose)
    ref_pose().aclref_local_to_object_space(_ptr(parents), _ptr(local_pose), local_pose.shape[0], _ptr(out))
    return out
