"""ctypes binding of the C ABI in include/aclhip.h (acl_amd/lib/libaclhip.so).

Mirrors the reference's decompression surface (acl::decompression_context::initialize / seek /
decompress_tracks / decompress_track, /root/reference/includes/acl/decompression/decompress.h:76-201)
for batches of clip instances. torch is used only for device memory and streams.

The HIP library is the only implementation: if it cannot be loaded this module raises, it never falls back to a CPU path.
"""
import ctypes
import os

import numpy as np

# ACLHIP_LIBRARY: another build of the same library (A/B measurements of kernel variants, see tools/)
_LIB_PATH = os.environ.get("ACLHIP_LIBRARY") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libaclhip.so")

ROUND_NONE, ROUND_FLOOR, ROUND_CEIL, ROUND_NEAREST, ROUND_PER_TRACK = 0, 1, 2, 3, 4
LOOP_CLAMP, LOOP_WRAP, LOOP_AS_COMPRESSED = 0, 1, 2
NORMALIZE_NEVER, NORMALIZE_LERP_ONLY, NORMALIZE_ALWAYS = 0, 1, 2
DEFAULT_SKIPPED, DEFAULT_CONSTANT, DEFAULT_VARIABLE, DEFAULT_LEGACY, DEFAULT_BIND_POSE = 0, 1, 2, 3, 4
# aclhip_status
(OK, ERROR_INVALID_ARGUMENT, ERROR_INVALID_CLIP, ERROR_UNSUPPORTED_FORMAT, ERROR_UNKNOWN_CLIP, ERROR_OUT_OF_MEMORY, ERROR_DEVICE, ERROR_NO_DEVICE,
 ERROR_UNKNOWN_DATABASE, ERROR_NOT_IN_DATABASE, ERROR_NO_METADATA) = range(11)
INVALID_HANDLE = 0xFFFFFFFF

EXPORTED_SYMBOLS = [
    "aclhip_status_string", "aclhip_last_error_message", "aclhip_abi_version", "aclhip_create", "aclhip_destroy", "aclhip_default_params",
    "aclhip_register_clip", "aclhip_unregister_clip", "aclhip_get_clip_info", "aclhip_clip_matches",
    "aclhip_decompress_tracks_batch", "aclhip_decompress_track_batch", "aclhip_decompress_tracks_host", "aclhip_decompress_track_host",
    "aclhip_get_rejected_instance_count", "aclhip_time_decompress_tracks_batch", "aclhip_batch_algorithmic_bytes",
    "aclhip_measure_write_bandwidth", "aclhip_measure_pose_store_bandwidth", "aclhip_describe_tracks_kernel",
    "aclhip_register_database", "aclhip_unregister_database", "aclhip_get_database_info", "aclhip_register_clip_with_database",
    "aclhip_database_stream_in", "aclhip_database_stream_out",
    "aclhip_all_gather_poses", "aclhip_probe_rccl", "aclhip_decompress_all_samples", "aclhip_check_clip", "aclhip_check_database",
    "aclhip_decompress_scalar_tracks_batch", "aclhip_decompress_scalar_track_batch", "aclhip_decompress_scalar_tracks_host", "aclhip_decompress_scalar_track_host",
    "aclhip_decompress_tracks_batch_rows", "aclhip_order_track_requests_for_locality", "aclhip_order_instances_for_locality", "aclhip_order_instances_device", "aclhip_order_instances_for_pose_windows",
    "aclhip_get_negative_scale_count", "aclhip_register_database_streamed", "aclhip_database_stream_in_from", "aclhip_get_lifetime_stats", "aclhip_peer_export_buffer", "aclhip_peer_open_buffer", "aclhip_peer_close_buffer", "aclhip_push_poses_to_peer",
    "aclhip_decompress_tracks_batch_out", "aclhip_decompress_tracks_host_out", "aclhip_layout_bytes_per_track",
    "aclhip_forget_stream", "aclhip_instance_list_create", "aclhip_instance_list_destroy", "aclhip_instance_list_set_clips", "aclhip_instance_list_update",
    "aclhip_decompress_tracks_list", "aclhip_instance_list_get_order", "aclhip_instance_list_attach", "aclhip_instance_list_note_changes",
    "aclhip_strip_database_tier", "aclhip_plan_hierarchy_walk", "aclhip_set_clip_hierarchy", "aclhip_decompress_poses_batch", "aclhip_decompress_poses_host", "aclhip_time_decompress_poses_batch",
    "aclhip_pose_windows_of_launch", "aclhip_order_instances_device_for_windows", "aclhip_describe_tracks_launch", "aclhip_analyze_clip",
    "aclhip_get_clip_metadata_info", "aclhip_get_clip_parent_indices", "aclhip_get_clip_track_descriptions", "aclhip_set_clip_hierarchy_from_metadata", "aclhip_read_clip_metadata",
]


class DecompressParams(ctypes.Structure):
    """aclhip_decompress_params"""
    _fields_ = [
        ("rounding_policy", ctypes.c_uint8), ("looping_policy", ctypes.c_uint8), ("normalization", ctypes.c_uint8), ("per_track_rounding", ctypes.c_uint8),
        ("default_rotation_mode", ctypes.c_uint8), ("default_translation_mode", ctypes.c_uint8), ("default_scale_mode", ctypes.c_uint8), ("reserved0", ctypes.c_uint8),
        ("default_values", ctypes.c_void_p), ("track_rounding_policies", ctypes.c_void_p), ("instance_rounding_policies", ctypes.c_void_p),
        ("instance_looping_policies", ctypes.c_void_p),
        ("track_rounding_table", ctypes.c_void_p), ("instance_rounding_tables", ctypes.c_void_p), ("track_rounding_stride", ctypes.c_uint32), ("flags", ctypes.c_uint32),
    ]


class PoseConsumers(ctypes.Structure):
    """aclhip_pose_consumers"""
    _fields_ = [
        ("additive_format", ctypes.c_uint32), ("object_space", ctypes.c_uint32),
        ("base_clips", ctypes.c_void_p), ("base_sample_times", ctypes.c_void_p), ("base_poses", ctypes.c_void_p), ("base_pose_stride_bytes", ctypes.c_uint64),
        ("num_blend_clips", ctypes.c_uint32), ("flags", ctypes.c_uint32),
        ("blend_clips", ctypes.c_void_p), ("blend_sample_times", ctypes.c_void_p), ("blend_weights", ctypes.c_void_p),
    ]


class OutputDesc(ctypes.Structure):
    """aclhip_output_desc"""
    _fields_ = [
        ("layout", ctypes.c_uint32), ("skip_rotations", ctypes.c_uint8), ("skip_translations", ctypes.c_uint8), ("skip_scales", ctypes.c_uint8), ("reserved0", ctypes.c_uint8),
        ("rows", ctypes.c_void_p), ("skip_tracks", ctypes.c_void_p),
        ("mask_table", ctypes.c_void_p), ("instance_masks", ctypes.c_void_p), ("instance_track_counts", ctypes.c_void_p), ("mask_stride", ctypes.c_uint32), ("reserved1", ctypes.c_uint32),
    ]


ABI_VERSION = 6             # ACLHIP_ABI_VERSION: the struct layouts mirrored above
PEER_HANDLE_BYTES = 72      # ACLHIP_PEER_HANDLE_BYTES
LAYOUT_QVV48, LAYOUT_QVV40, LAYOUT_QV32 = 0, 1, 2  # aclhip_pose_layout
LAYOUTS = {"qvv48": (LAYOUT_QVV48, 48), "qvv40": (LAYOUT_QVV40, 40), "qv32": (LAYOUT_QV32, 32)}     # name -> (aclhip_pose_layout, bytes per track)


def relayout_pose(pose, layout, skip=(False, False, False), into=None):
    """Test helper (numpy, host): what a [..., num_tracks, 12] QVV48 pose looks like through `layout` with the sub-track kinds of
    `skip` (rotation, translation, scale) left untouched. Returns [..., num_tracks, floats per track]; skipped pieces keep the
    values of `into` (zeros when it is not given)."""
    pose = np.asarray(pose, dtype=np.float32)
    columns = {LAYOUT_QVV48: ([0, 1, 2, 3], [4, 5, 6, 7], [8, 9, 10, 11]), LAYOUT_QVV40: ([0, 1, 2, 3], [4, 5, 6], [7, 8, 9]), LAYOUT_QV32: ([0, 1, 2, 3], [4, 5, 6, 7], [])}[layout]
    sources = ([0, 1, 2, 3], [4, 5, 6, 7], [8, 9, 10, 11])
    width = {LAYOUT_QVV48: 12, LAYOUT_QVV40: 10, LAYOUT_QV32: 8}[layout]
    out = np.zeros(pose.shape[:-1] + (width,), dtype=np.float32) if into is None else np.array(into, dtype=np.float32, copy=True)
    for kind in range(3):
        if skip[kind] or not columns[kind]:
            continue
        out[..., columns[kind]] = pose[..., sources[kind][: len(columns[kind])]]
    return out


ADDITIVE_NONE, ADDITIVE_RELATIVE, ADDITIVE_ADDITIVE0, ADDITIVE_ADDITIVE1 = 0, 1, 2, 3  # aclhip_additive_format
CONSUMERS_FAST = 1          # ACLHIP_CONSUMERS_FAST (aclhip_pose_consumers::flags)
DECODE_FAST = 1             # ACLHIP_DECODE_FAST (aclhip_decompress_params::flags)
NO_PARENT = 0xFFFFFFFF


class ClipInfo(ctypes.Structure):
    """aclhip_clip_info"""
    _fields_ = [
        ("num_tracks", ctypes.c_uint32), ("num_samples", ctypes.c_uint32), ("sample_rate", ctypes.c_float), ("duration", ctypes.c_float),
        ("num_segments", ctypes.c_uint32), ("has_scale", ctypes.c_uint32), ("looping_policy", ctypes.c_uint32), ("compressed_size", ctypes.c_uint32),
        ("hash", ctypes.c_uint32), ("num_animated_sub_tracks", ctypes.c_uint32), ("has_database", ctypes.c_uint32), ("has_stripped_keyframes", ctypes.c_uint32),
        ("track_type", ctypes.c_uint32), ("num_components", ctypes.c_uint32),
    ]


class ClipMetadataInfo(ctypes.Structure):
    """aclhip_clip_metadata_info"""
    _fields_ = [("has_metadata", ctypes.c_uint32), ("has_parent_track_indices", ctypes.c_uint32), ("has_track_descriptions", ctypes.c_uint32),
                ("has_track_names", ctypes.c_uint32), ("has_track_list_name", ctypes.c_uint32), ("has_contributing_error", ctypes.c_uint32)]


class DatabaseInfo(ctypes.Structure):
    """aclhip_database_info"""
    _fields_ = [
        ("num_clips", ctypes.c_uint32), ("num_segments", ctypes.c_uint32), ("max_chunk_size", ctypes.c_uint32),
        ("num_chunks", ctypes.c_uint32 * 2), ("num_loaded_chunks", ctypes.c_uint32 * 2), ("bulk_data_size", ctypes.c_uint32 * 2),
    ]


TIER_MEDIUM_IMPORTANCE = 1  # quality_tier::medium_importance (core/quality_tiers.h)
TIER_LOWEST_IMPORTANCE = 2


class AclHipError(RuntimeError):
    def __init__(self, status, message):
        super().__init__(f"aclhip status {status}: {message}")
        self.status = status


_lib = None


def library_path():
    return _LIB_PATH


def load_library():
    """Loads libaclhip.so (raises when it was not built -- there is no fallback).

    PyTorch is imported first when it is installed: it bundles its own HIP runtime and the two cannot be loaded in the other order."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIB_PATH):
        raise RuntimeError(f"{_LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'`; there is no CPU fallback")
    # Load order: this Python layer hands torch tensors' device pointers to the library, and the PyTorch wheel bundles its own HIP
    # runtime. Whichever of the two runtimes is loaded second finds "no HIP device", so torch (when present) always goes first.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = ctypes.CDLL(_LIB_PATH)
    vp, u32, u64, i32 = ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint64, ctypes.c_int
    pparams = ctypes.POINTER(DecompressParams)
    lib.aclhip_status_string.argtypes = [i32]
    lib.aclhip_status_string.restype = ctypes.c_char_p
    lib.aclhip_last_error_message.argtypes = [vp]
    lib.aclhip_last_error_message.restype = ctypes.c_char_p
    lib.aclhip_abi_version.argtypes = []
    lib.aclhip_abi_version.restype = ctypes.c_uint32
    if lib.aclhip_abi_version() != ABI_VERSION:
        raise RuntimeError(f"{_LIB_PATH} was built with ABI version {lib.aclhip_abi_version()}, this binding mirrors version {ABI_VERSION}: rebuild the library")
    lib.aclhip_create.argtypes = [i32, ctypes.POINTER(vp)]
    lib.aclhip_destroy.argtypes = [vp]
    lib.aclhip_destroy.restype = None
    lib.aclhip_default_params.argtypes = [pparams]
    lib.aclhip_default_params.restype = None
    lib.aclhip_register_clip.argtypes = [vp, vp, u64, i32, ctypes.POINTER(u32)]
    lib.aclhip_unregister_clip.argtypes = [vp, u32]
    lib.aclhip_get_clip_info.argtypes = [vp, u32, ctypes.POINTER(ClipInfo)]
    lib.aclhip_clip_matches.argtypes = [vp, u32, vp, ctypes.POINTER(i32)]
    lib.aclhip_decompress_tracks_batch.argtypes = [vp, vp, vp, u32, pparams, vp, u64, vp]
    lib.aclhip_decompress_track_batch.argtypes = [vp, vp, vp, vp, u32, pparams, vp, vp]
    lib.aclhip_decompress_tracks_host.argtypes = [vp, vp, vp, u32, pparams, u32, vp, u64]
    lib.aclhip_decompress_track_host.argtypes = [vp, vp, vp, vp, u32, pparams, u32, vp]
    lib.aclhip_get_rejected_instance_count.argtypes = [vp, ctypes.POINTER(u64)]
    lib.aclhip_time_decompress_tracks_batch.argtypes = [vp, vp, vp, u32, pparams, vp, u64, vp, u32, ctypes.POINTER(ctypes.c_float)]
    lib.aclhip_measure_write_bandwidth.argtypes = [vp, vp, u64, u32, vp, ctypes.POINTER(ctypes.c_float)]
    lib.aclhip_describe_tracks_kernel.argtypes = [vp, pparams, ctypes.c_char_p, u32]
    lib.aclhip_batch_algorithmic_bytes.argtypes = [vp, vp, u32, ctypes.POINTER(u64), ctypes.POINTER(u64)]
    lib.aclhip_register_database.argtypes = [vp, vp, u64, vp, vp, i32, ctypes.POINTER(u32)]
    lib.aclhip_unregister_database.argtypes = [vp, u32]
    lib.aclhip_get_database_info.argtypes = [vp, u32, ctypes.POINTER(DatabaseInfo)]
    lib.aclhip_register_clip_with_database.argtypes = [vp, vp, u64, i32, u32, ctypes.POINTER(u32)]
    lib.aclhip_database_stream_in.argtypes = [vp, u32, u32, u32, vp, ctypes.POINTER(u32)]
    lib.aclhip_database_stream_out.argtypes = [vp, u32, u32, u32, vp, ctypes.POINTER(u32)]
    lib.aclhip_check_clip.argtypes = [vp, u64, i32, ctypes.c_char_p, u32]
    lib.aclhip_analyze_clip.argtypes = [vp, u64, i32, ctypes.POINTER(ctypes.c_uint32)]
    lib.aclhip_check_database.argtypes = [vp, u64, vp, vp, i32, ctypes.c_char_p, u32]
    lib.aclhip_decompress_all_samples.argtypes = [vp, u32, pparams, vp, vp, u64, vp]
    lib.aclhip_all_gather_poses.argtypes = [vp, vp, vp, vp, u64, vp]
    lib.aclhip_decompress_scalar_tracks_batch.argtypes = [vp, vp, vp, u32, pparams, vp, u64, vp]
    lib.aclhip_decompress_scalar_track_batch.argtypes = [vp, vp, vp, vp, u32, pparams, vp, u64, vp]
    lib.aclhip_decompress_scalar_tracks_host.argtypes = [vp, vp, vp, u32, pparams, vp, u64]
    lib.aclhip_decompress_scalar_track_host.argtypes = [vp, vp, vp, vp, u32, pparams, vp, u64]
    lib.aclhip_decompress_tracks_batch_rows.argtypes = [vp, vp, vp, vp, u32, pparams, vp, u64, vp]
    lib.aclhip_order_instances_for_locality.argtypes = [vp, vp, u32, vp]
    lib.aclhip_order_instances_for_pose_windows.argtypes = [u32, vp, u32, vp]
    lib.aclhip_order_track_requests_for_locality.argtypes = [vp, u32, vp]
    lib.aclhip_order_instances_device.argtypes = [vp, vp, vp, u32, vp, vp, vp, vp]
    pconsumers = ctypes.POINTER(PoseConsumers)
    poutput = ctypes.POINTER(OutputDesc)
    lib.aclhip_get_lifetime_stats.argtypes = [vp, ctypes.POINTER(u64)]
    lib.aclhip_get_negative_scale_count.argtypes = [vp, ctypes.POINTER(u64)]
    lib.aclhip_register_database_streamed.argtypes = [vp, vp, u64, i32, ctypes.POINTER(u32)]
    lib.aclhip_database_stream_in_from.argtypes = [vp, u32, u32, u32, vp, vp, ctypes.POINTER(u32)]
    lib.aclhip_peer_export_buffer.argtypes = [vp, vp, vp]
    lib.aclhip_peer_open_buffer.argtypes = [vp, vp, ctypes.POINTER(vp)]
    lib.aclhip_peer_close_buffer.argtypes = [vp, vp]
    lib.aclhip_push_poses_to_peer.argtypes = [vp, vp, u64, vp, u64, vp]
    lib.aclhip_decompress_tracks_batch_out.argtypes = [vp, vp, vp, u32, pparams, poutput, vp, u64, vp]
    lib.aclhip_decompress_tracks_host_out.argtypes = [vp, vp, vp, u32, pparams, u32, poutput, vp, u64]
    lib.aclhip_layout_bytes_per_track.argtypes = [u32]
    lib.aclhip_layout_bytes_per_track.restype = u32
    lib.aclhip_strip_database_tier.argtypes = [vp, u64, u32, vp, u64, ctypes.POINTER(u64)]
    lib.aclhip_plan_hierarchy_walk.argtypes = [vp, u32, u32, vp, ctypes.POINTER(u32)]
    lib.aclhip_set_clip_hierarchy.argtypes = [vp, u32, vp, u32]
    lib.aclhip_decompress_poses_batch.argtypes = [vp, vp, vp, u32, pparams, pconsumers, vp, u64, vp]
    lib.aclhip_decompress_poses_host.argtypes = [vp, vp, vp, u32, pparams, pconsumers, vp, u64]
    lib.aclhip_time_decompress_poses_batch.argtypes = [vp, vp, vp, u32, pparams, pconsumers, vp, u64, vp, u32, ctypes.POINTER(ctypes.c_float)]
    lib.aclhip_pose_windows_of_launch.argtypes = [vp, u32, u64, ctypes.POINTER(u32)]
    lib.aclhip_order_instances_device_for_windows.argtypes = [vp, u32, vp, vp, u32, vp, vp, vp, vp]
    lib.aclhip_describe_tracks_launch.argtypes = [vp, pparams, poutput, u64, ctypes.c_char_p, u32, ctypes.POINTER(u32)]
    lib.aclhip_get_clip_metadata_info.argtypes = [vp, u32, ctypes.POINTER(ClipMetadataInfo)]
    lib.aclhip_get_clip_parent_indices.argtypes = [vp, u32, vp, u32]
    lib.aclhip_get_clip_track_descriptions.argtypes = [vp, u32, vp, vp, vp, u32]
    lib.aclhip_set_clip_hierarchy_from_metadata.argtypes = [vp, u32]
    lib.aclhip_read_clip_metadata.argtypes = [vp, u64, ctypes.POINTER(ClipMetadataInfo), vp, vp, vp, vp, u32]
    _lib = lib
    return lib


def probe_rccl():
    """aclhip_probe_rccl: (version code, path of the library, how it was found) of the RCCL aclhip_all_gather_poses would call; raises AclHipError when there is none"""
    lib = load_library()
    version, path, how = ctypes.c_int(0), ctypes.create_string_buffer(512), ctypes.create_string_buffer(128)
    status = lib.aclhip_probe_rccl(ctypes.byref(version), path, 512, how, 128)
    if status != 0:
        raise AclHipError(status, lib.aclhip_last_error_message(None).decode())
    return version.value, path.value.decode(), how.value.decode()


def check_clip(blob, check_hash=True):
    """Host only validation of a compressed_tracks blob (no GPU needed). Returns (status, message); status 0 = valid."""
    array = np.frombuffer(blob, dtype=np.uint8) if not isinstance(blob, np.ndarray) else blob
    message = ctypes.create_string_buffer(512)
    status = load_library().aclhip_check_clip(array.ctypes.data, array.size, 1 if check_hash else 0, message, 512)
    return status, message.value.decode()


CLIP_FACT_SHORT_EXACT_MATH, CLIP_FACT_RAW_ROTATIONS, CLIP_FACT_NEGATIVE_SCALE = 1, 2, 4


def read_clip_metadata(blob):
    """aclhip_read_clip_metadata (host only): (ClipMetadataInfo, parents or None, (default_values [n, 12], precisions, shell_distances) or None)"""
    lib = load_library()
    num_tracks = int(np.frombuffer(bytes(blob[16:20]), dtype=np.uint32)[0])
    info = ClipMetadataInfo()
    parents = np.zeros(max(num_tracks, 1), dtype=np.uint32)
    defaults, precisions, shells = np.zeros((max(num_tracks, 1), 12), dtype=np.float32), np.zeros(max(num_tracks, 1), dtype=np.float32), np.zeros(max(num_tracks, 1), dtype=np.float32)
    status = lib.aclhip_read_clip_metadata(blob.ctypes.data, blob.size, ctypes.byref(info), parents.ctypes.data, defaults.ctypes.data, precisions.ctypes.data, shells.ctypes.data, max(num_tracks, 1))
    if status != 0:
        raise AclHipError(status, lib.aclhip_last_error_message(None).decode())
    return (info, parents[:num_tracks] if info.has_parent_track_indices else None,
            (defaults[:num_tracks], precisions[:num_tracks], shells[:num_tracks]) if info.has_track_descriptions else None)


def analyze_clip(blob, check_hash=True):
    """Host only: what registration derives about the values a clip can decode to (aclhip_analyze_clip). Returns the CLIP_FACT_* bits."""
    array = np.frombuffer(blob, dtype=np.uint8) if not isinstance(blob, np.ndarray) else blob
    facts = ctypes.c_uint32(0)
    status = load_library().aclhip_analyze_clip(array.ctypes.data, array.size, 1 if check_hash else 0, ctypes.byref(facts))
    if status != 0:
        raise AclHipError(status, "aclhip_analyze_clip")
    return int(facts.value)


def check_database(database, bulk_data_medium=None, bulk_data_low=None, check_hash=True):
    """Host only validation of a compressed_database (+ split bulk data). Returns (status, message)."""
    as_array = lambda b: None if b is None else (np.frombuffer(b, dtype=np.uint8) if not isinstance(b, np.ndarray) else b)
    array, medium, low = as_array(database), as_array(bulk_data_medium), as_array(bulk_data_low)
    message = ctypes.create_string_buffer(512)
    status = load_library().aclhip_check_database(array.ctypes.data, array.size, _host_ptr(medium), _host_ptr(low), 1 if check_hash else 0, message, 512)
    return status, message.value.decode()


def order_instances_for_locality(clips, windows_per_instance=1):
    """aclhip_order_instances_for_pose_windows (no context: the caller says how many wavefronts a pose takes): host only, no GPU needed."""
    clips = np.ascontiguousarray(clips, dtype=np.uint32)
    order = np.empty(clips.size, dtype=np.uint32)
    status = load_library().aclhip_order_instances_for_pose_windows(windows_per_instance, clips.ctypes.data, clips.size, order.ctypes.data)
    if status != 0:
        raise AclHipError(status, "aclhip_order_instances_for_pose_windows failed")
    return order


def order_track_requests_for_locality(clips):
    """aclhip_order_track_requests_for_locality: host only, no GPU needed."""
    clips = np.ascontiguousarray(clips, dtype=np.uint32)
    order = np.empty(clips.size, dtype=np.uint32)
    status = load_library().aclhip_order_track_requests_for_locality(clips.ctypes.data, clips.size, order.ctypes.data)
    if status != 0:
        raise AclHipError(status, "aclhip_order_track_requests_for_locality failed")
    return order


def strip_database_tier(database, tier):
    """aclhip_strip_database_tier (host only): (status, stripped compressed_database as a 16 byte aligned uint8 array or None)"""
    from .synth import aligned_bytes
    array = np.frombuffer(database, dtype=np.uint8) if not isinstance(database, np.ndarray) else database
    lib = load_library()
    size = ctypes.c_uint64(0)
    status = lib.aclhip_strip_database_tier(array.ctypes.data, array.size, int(tier), None, 0, ctypes.byref(size))
    if status != 0:
        return status, None
    out = aligned_bytes(size.value)
    status = lib.aclhip_strip_database_tier(array.ctypes.data, array.size, int(tier), out.ctypes.data, out.size, ctypes.byref(size))
    return status, (out if status == 0 else None)


def plan_hierarchy_walk(parent_indices, transforms_per_step):
    """aclhip_plan_hierarchy_walk (host only): (number of steps, 1-based step per transform with 0 for roots)"""
    parents = np.ascontiguousarray(parent_indices, dtype=np.uint32)
    steps = np.zeros(parents.size, dtype=np.uint32)
    num_steps = ctypes.c_uint32(0)
    status = load_library().aclhip_plan_hierarchy_walk(parents.ctypes.data, parents.size, int(transforms_per_step), steps.ctypes.data, ctypes.byref(num_steps))
    if status != 0:
        raise AclHipError(status, "aclhip_plan_hierarchy_walk: transforms must be sorted parent first")
    return num_steps.value, steps


def default_params(**overrides):
    params = DecompressParams()
    load_library().aclhip_default_params(ctypes.byref(params))
    for key, value in overrides.items():
        if not hasattr(params, key):
            raise AttributeError(f"aclhip_decompress_params has no field '{key}'")
        setattr(params, key, value)
    return params


def _host_ptr(array):
    return array.ctypes.data if array is not None else None


class Context:
    """An aclhip_context bound to one HIP device: owns the HBM copies of registered clips."""

    def __init__(self, device_index=0):
        self._lib = load_library()
        handle = ctypes.c_void_p()
        status = self._lib.aclhip_create(device_index, ctypes.byref(handle))
        if status != 0:
            raise AclHipError(status, self._lib.aclhip_status_string(status).decode())
        self._handle = handle
        self.device_index = device_index

    def close(self):
        if getattr(self, "_handle", None):
            self._lib.aclhip_destroy(self._handle)
            self._handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def _check(self, status):
        if status != 0:
            message = self._lib.aclhip_last_error_message(self._handle).decode() or self._lib.aclhip_status_string(status).decode()
            raise AclHipError(status, message)

    # ---- clips (decompression_context::initialize) ----
    def register_clip(self, blob, check_hash=True):
        """blob: bytes-like / uint8 numpy array holding one compressed_tracks. Returns the clip handle."""
        array = np.frombuffer(blob, dtype=np.uint8) if not isinstance(blob, np.ndarray) else blob
        handle = ctypes.c_uint32(INVALID_HANDLE)
        self._check(self._lib.aclhip_register_clip(self._handle, array.ctypes.data, array.size, 1 if check_hash else 0, ctypes.byref(handle)))
        return handle.value

    def unregister_clip(self, clip):
        self._check(self._lib.aclhip_unregister_clip(self._handle, clip))

    # ---- persistent instance lists (aclhip_instance_list_*): device pointers, stream ordered ----
    def instance_list_create(self, num_instances):
        handle = ctypes.c_uint32(INVALID_HANDLE)
        self._check(self._lib.aclhip_instance_list_create(self._handle, ctypes.c_uint32(num_instances), ctypes.byref(handle)))
        return handle.value

    def instance_list_destroy(self, instance_list):
        self._check(self._lib.aclhip_instance_list_destroy(self._handle, ctypes.c_uint32(instance_list)))

    def instance_list_set_clips(self, instance_list, clips_ptr, stream=None):
        self._check(self._lib.aclhip_instance_list_set_clips(self._handle, ctypes.c_uint32(instance_list), ctypes.c_void_p(clips_ptr), ctypes.c_void_p(stream)))

    def instance_list_update(self, instance_list, instances_ptr, clips_ptr, count, stream=None):
        self._check(self._lib.aclhip_instance_list_update(self._handle, ctypes.c_uint32(instance_list), ctypes.c_void_p(instances_ptr), ctypes.c_void_p(clips_ptr),
                                                          ctypes.c_uint32(count), ctypes.c_void_p(stream)))

    def instance_list_attach(self, instance_list, caller_clips_ptr, stream=None):
        """aclhip_instance_list_attach: the list decodes the caller's own device clip array (kept alive and in place by the caller)"""
        self._check(self._lib.aclhip_instance_list_attach(self._handle, ctypes.c_uint32(instance_list), ctypes.c_void_p(caller_clips_ptr), ctypes.c_void_p(stream)))

    def instance_list_note_changes(self, instance_list, count):
        self._check(self._lib.aclhip_instance_list_note_changes(self._handle, ctypes.c_uint32(instance_list), ctypes.c_uint32(count)))

    def decompress_tracks_list(self, instance_list, times_ptr, poses_ptr, pose_stride_bytes, params=None, output=None, poses_in_instance_order=False, stream=None):
        params = params if params is not None else default_params()
        self._check(self._lib.aclhip_decompress_tracks_list(self._handle, ctypes.c_uint32(instance_list), ctypes.c_void_p(times_ptr), ctypes.byref(params),
                                                            ctypes.byref(output) if output is not None else None, ctypes.c_int(1 if poses_in_instance_order else 0),
                                                            ctypes.c_void_p(poses_ptr), ctypes.c_uint64(pose_stride_bytes), ctypes.c_void_p(stream)))

    def instance_list_order(self, instance_list):
        """(device address of the slot -> instance order, number of orderings so far)"""
        order, count = ctypes.c_void_p(0), ctypes.c_uint64(0)
        self._check(self._lib.aclhip_instance_list_get_order(self._handle, ctypes.c_uint32(instance_list), ctypes.byref(order), ctypes.byref(count)))
        return order.value, count.value

    def forget_stream(self, stream):
        """aclhip_forget_stream: before destroying a stream the context has launched on"""
        self._check(self._lib.aclhip_forget_stream(self._handle, ctypes.c_void_p(stream)))

    def clip_info(self, clip):
        info = ClipInfo()
        self._check(self._lib.aclhip_get_clip_info(self._handle, clip, ctypes.byref(info)))
        return info

    def clip_metadata_info(self, clip):
        info = ClipMetadataInfo()
        self._check(self._lib.aclhip_get_clip_metadata_info(self._handle, clip, ctypes.byref(info)))
        return info

    def clip_parent_indices(self, clip):
        """compressed_tracks::get_parent_track_index of every track (NO_PARENT for roots); AclHipError(ERROR_NO_METADATA) when not stored"""
        parents = np.zeros(self.clip_info(clip).num_tracks, dtype=np.uint32)
        self._check(self._lib.aclhip_get_clip_parent_indices(self._handle, clip, parents.ctypes.data, parents.size))
        return parents

    def clip_track_descriptions(self, clip):
        """(default_values [num_tracks, 12], precisions, shell_distances) from the blob's track descriptions"""
        num_tracks = self.clip_info(clip).num_tracks
        defaults, precisions, shells = np.zeros((num_tracks, 12), dtype=np.float32), np.zeros(num_tracks, dtype=np.float32), np.zeros(num_tracks, dtype=np.float32)
        self._check(self._lib.aclhip_get_clip_track_descriptions(self._handle, clip, defaults.ctypes.data, precisions.ctypes.data, shells.ctypes.data, num_tracks))
        return defaults, precisions, shells

    def set_clip_hierarchy_from_metadata(self, clip):
        self._check(self._lib.aclhip_set_clip_hierarchy_from_metadata(self._handle, clip))

    def clip_matches(self, clip, blob):
        matches = ctypes.c_int(0)
        self._check(self._lib.aclhip_clip_matches(self._handle, clip, blob.ctypes.data, ctypes.byref(matches)))
        return bool(matches.value)

    # ---- databases (database_context) ----
    def register_database(self, database, bulk_data_medium=None, bulk_data_low=None, check_hash=True):
        """database: one compressed_database (bytes-like / uint8 array); bulk data arrays when it was split off. Returns the handle."""
        as_array = lambda b: None if b is None else (np.frombuffer(b, dtype=np.uint8) if not isinstance(b, np.ndarray) else b)
        array, medium, low = as_array(database), as_array(bulk_data_medium), as_array(bulk_data_low)
        handle = ctypes.c_uint32(INVALID_HANDLE)
        self._check(self._lib.aclhip_register_database(self._handle, array.ctypes.data, array.size, _host_ptr(medium), _host_ptr(low),
                                                       1 if check_hash else 0, ctypes.byref(handle)))
        return handle.value

    def register_database_streamed(self, database, check_hash=True):
        """aclhip_register_database_streamed: the bulk data arrives with the stream-in requests (database_stream_in_from)"""
        handle = ctypes.c_uint32(INVALID_HANDLE)
        self._check(self._lib.aclhip_register_database_streamed(self._handle, database.ctypes.data, database.size, int(check_hash), ctypes.byref(handle)))
        return handle.value

    def database_stream_in_from(self, database, tier, tier_bulk_data, num_chunks=0xFFFFFFFF, stream=None):
        moved = ctypes.c_uint32(0)
        self._check(self._lib.aclhip_database_stream_in_from(self._handle, database, tier, num_chunks, tier_bulk_data.ctypes.data, stream, ctypes.byref(moved)))
        return moved.value

    def unregister_database(self, database):
        self._check(self._lib.aclhip_unregister_database(self._handle, database))

    def database_info(self, database):
        info = DatabaseInfo()
        self._check(self._lib.aclhip_get_database_info(self._handle, database, ctypes.byref(info)))
        return info

    def register_clip_with_database(self, blob, database, check_hash=True):
        array = np.frombuffer(blob, dtype=np.uint8) if not isinstance(blob, np.ndarray) else blob
        handle = ctypes.c_uint32(INVALID_HANDLE)
        self._check(self._lib.aclhip_register_clip_with_database(self._handle, array.ctypes.data, array.size, 1 if check_hash else 0, database, ctypes.byref(handle)))
        return handle.value

    def database_stream_in(self, database, tier, num_chunks=0xFFFFFFFF, stream=None):
        """database_context::stream_in(tier, num_chunks); returns how many chunks were enqueued (0 = nothing left)."""
        moved = ctypes.c_uint32(0)
        self._check(self._lib.aclhip_database_stream_in(self._handle, database, tier, num_chunks, stream, ctypes.byref(moved)))
        return moved.value

    def database_stream_out(self, database, tier, num_chunks=0xFFFFFFFF, stream=None):
        moved = ctypes.c_uint32(0)
        self._check(self._lib.aclhip_database_stream_out(self._handle, database, tier, num_chunks, stream, ctypes.byref(moved)))
        return moved.value

    # ---- device pointer API (inputs and outputs resident in HBM) ----
    def decompress_tracks_batch(self, clips_ptr, times_ptr, num_instances, poses_ptr, pose_stride_bytes, params=None, stream=None):
        """seek + decompress_tracks for every instance. All pointers are device addresses (ints)."""
        params = params if params is not None else default_params()
        self._check(self._lib.aclhip_decompress_tracks_batch(self._handle, clips_ptr, times_ptr, num_instances, ctypes.byref(params), poses_ptr, pose_stride_bytes, stream))

    def decompress_tracks_batch_rows(self, clips_ptr, times_ptr, rows_ptr, num_instances, poses_ptr, pose_stride_bytes, params=None, stream=None):
        """aclhip_decompress_tracks_batch with the pose of instance i stored at row rows[i] (device array)."""
        params = params if params is not None else default_params()
        self._check(self._lib.aclhip_decompress_tracks_batch_rows(self._handle, clips_ptr, times_ptr, rows_ptr, num_instances, ctypes.byref(params), poses_ptr, pose_stride_bytes, stream))

    def decompress_tracks_batch_out(self, clips_ptr, times_ptr, num_instances, poses_ptr, pose_stride_bytes, output, params=None, stream=None):
        """aclhip_decompress_tracks_batch with an OutputDesc (layout, skipped sub-track kinds, rows)."""
        params = params if params is not None else default_params()
        self._check(self._lib.aclhip_decompress_tracks_batch_out(self._handle, clips_ptr, times_ptr, num_instances, ctypes.byref(params), ctypes.byref(output), poses_ptr, pose_stride_bytes, stream))

    def order_instances_for_locality(self, clips):
        """Host only: the permutation aclhip_order_instances_for_locality computes for the instance list `clips`."""
        clips = np.ascontiguousarray(clips, dtype=np.uint32)
        order = np.empty(clips.size, dtype=np.uint32)
        self._check(self._lib.aclhip_order_instances_for_locality(self._handle, clips.ctypes.data, clips.size, order.ctypes.data))
        return order

    def order_instances_device(self, clips_ptr, times_ptr, num_instances, order_ptr, out_clips_ptr=None, out_times_ptr=None, stream=None):
        """aclhip_order_instances_device: the locality order of a DEVICE instance list, stream ordered (device pointers)."""
        self._check(self._lib.aclhip_order_instances_device(self._handle, clips_ptr, times_ptr, num_instances, order_ptr, out_clips_ptr, out_times_ptr, stream))

    def order_instances_device_for_windows(self, windows_per_instance, clips_ptr, times_ptr, num_instances, order_ptr, out_clips_ptr=None, out_times_ptr=None, stream=None):
        """aclhip_order_instances_device_for_windows: the same for launches of `windows_per_instance` wavefronts per pose (pose_windows_of_launch)."""
        self._check(self._lib.aclhip_order_instances_device_for_windows(self._handle, windows_per_instance, clips_ptr, times_ptr, num_instances, order_ptr, out_clips_ptr, out_times_ptr, stream))

    def pose_windows_of_launch(self, pose_stride_bytes, layout=LAYOUT_QVV48):
        """Wavefronts per instance of a pose launch with rows of `pose_stride_bytes` and the clips registered now."""
        windows = ctypes.c_uint32(0)
        self._check(self._lib.aclhip_pose_windows_of_launch(self._handle, int(layout), int(pose_stride_bytes), ctypes.byref(windows)))
        return windows.value

    def decompress_track_batch(self, clips_ptr, times_ptr, tracks_ptr, num_instances, out_ptr, params=None, stream=None):
        params = params if params is not None else default_params()
        self._check(self._lib.aclhip_decompress_track_batch(self._handle, clips_ptr, times_ptr, tracks_ptr, num_instances, ctypes.byref(params), out_ptr, stream))

    def time_decompress_tracks_batch(self, clips_ptr, times_ptr, num_instances, poses_ptr, pose_stride_bytes, repeats, params=None, stream=None):
        """Average device milliseconds per launch, HIP events recorded on `stream`."""
        params = params if params is not None else default_params()
        ms = ctypes.c_float(0.0)
        self._check(self._lib.aclhip_time_decompress_tracks_batch(self._handle, clips_ptr, times_ptr, num_instances, ctypes.byref(params), poses_ptr, pose_stride_bytes, stream, repeats, ctypes.byref(ms)))
        return ms.value

    # ---- pose consumers: additive apply and local -> object space fused into the decode ----
    def set_clip_hierarchy(self, clip, parent_indices):
        parents = np.ascontiguousarray(parent_indices, dtype=np.uint32)
        self._check(self._lib.aclhip_set_clip_hierarchy(self._handle, clip, parents.ctypes.data, parents.size))

    def decompress_poses_batch(self, clips_ptr, times_ptr, num_instances, poses_ptr, pose_stride_bytes, consumers, params=None, stream=None):
        """aclhip_decompress_poses_batch; `consumers` is a PoseConsumers holding device addresses."""
        params = params if params is not None else default_params()
        self._check(self._lib.aclhip_decompress_poses_batch(self._handle, clips_ptr, times_ptr, num_instances, ctypes.byref(params), ctypes.byref(consumers), poses_ptr, pose_stride_bytes, stream))

    def time_decompress_poses_batch(self, clips_ptr, times_ptr, num_instances, poses_ptr, pose_stride_bytes, consumers, repeats, params=None, stream=None):
        params = params if params is not None else default_params()
        ms = ctypes.c_float(0.0)
        self._check(self._lib.aclhip_time_decompress_poses_batch(self._handle, clips_ptr, times_ptr, num_instances, ctypes.byref(params), ctypes.byref(consumers), poses_ptr, pose_stride_bytes, stream, repeats, ctypes.byref(ms)))
        return ms.value

    def decompress_poses(self, clips, sample_times, additive_format=ADDITIVE_NONE, object_space=False, base_clips=None, base_sample_times=None, base_poses=None,
                         params=None, num_tracks=None, out=None, instance_rounding=None, blend_clips=None, blend_sample_times=None, blend_weights=None, flags=0):
        """Host arrays in, host poses out: float32 [n, num_tracks, 12] after the consumers. The base of an additive instance is either
        (base_clips[i], base_sample_times[i]) or base_poses[i] ([n, num_tracks, 12]). A blend of K clips per instance: blend_clips /
        blend_sample_times [n, K - 1] (the further clips), blend_weights [n, K]."""
        clips = np.ascontiguousarray(clips, dtype=np.uint32)
        sample_times = np.ascontiguousarray(sample_times, dtype=np.float32)
        n = clips.size
        if num_tracks is None:
            num_tracks = max((self.clip_info(int(c)).num_tracks for c in np.unique(clips)), default=0)
        if out is None:
            out = np.zeros((n, num_tracks, 12), dtype=np.float32)
        params = params if params is not None else default_params()
        if instance_rounding is not None:
            instance_rounding = np.ascontiguousarray(instance_rounding, dtype=np.uint8)
            params.instance_rounding_policies = instance_rounding.ctypes.data
        consumers = PoseConsumers()
        consumers.additive_format = int(additive_format)
        consumers.object_space = 1 if object_space else 0
        consumers.flags = int(flags)
        if base_clips is not None:
            base_clips = np.ascontiguousarray(base_clips, dtype=np.uint32)
            base_sample_times = np.ascontiguousarray(base_sample_times, dtype=np.float32)
            consumers.base_clips = base_clips.ctypes.data
            consumers.base_sample_times = base_sample_times.ctypes.data
        if base_poses is not None:
            base_poses = np.ascontiguousarray(base_poses, dtype=np.float32)
            consumers.base_poses = base_poses.ctypes.data
            consumers.base_pose_stride_bytes = base_poses.strides[0] if base_poses.ndim == 3 else num_tracks * 48
        if blend_weights is not None:
            blend_weights = np.ascontiguousarray(blend_weights, dtype=np.float32).reshape(n, -1)
            blend_clips = np.ascontiguousarray(blend_clips, dtype=np.uint32).reshape(n, -1)
            blend_sample_times = np.ascontiguousarray(blend_sample_times, dtype=np.float32).reshape(n, -1)
            consumers.num_blend_clips = blend_weights.shape[1]
            consumers.blend_clips = blend_clips.ctypes.data
            consumers.blend_sample_times = blend_sample_times.ctypes.data
            consumers.blend_weights = blend_weights.ctypes.data
        self._check(self._lib.aclhip_decompress_poses_host(self._handle, clips.ctypes.data, sample_times.ctypes.data, n, ctypes.byref(params), ctypes.byref(consumers), out.ctypes.data, num_tracks * 48))
        return out

    # ---- host pointer convenience API ----
    def decompress_tracks(self, clips, sample_times, params=None, num_tracks=None, out=None, default_values=None, track_rounding=None, instance_rounding=None):
        """Host arrays in, host poses out: returns float32 [n, num_tracks, 12]."""
        clips = np.ascontiguousarray(clips, dtype=np.uint32)
        sample_times = np.ascontiguousarray(sample_times, dtype=np.float32)
        n = clips.size
        if num_tracks is None:
            num_tracks = max((self.clip_info(int(c)).num_tracks for c in np.unique(clips)), default=0)
        if out is None:
            out = np.zeros((n, num_tracks, 12), dtype=np.float32)
        params = params if params is not None else default_params()
        count = 0
        if default_values is not None:
            default_values = np.ascontiguousarray(default_values, dtype=np.float32)
            params.default_values = default_values.ctypes.data
            count = default_values.size // 12
        if track_rounding is not None:
            track_rounding = np.ascontiguousarray(track_rounding, dtype=np.uint8)
            params.track_rounding_policies = track_rounding.ctypes.data
        if instance_rounding is not None:
            instance_rounding = np.ascontiguousarray(instance_rounding, dtype=np.uint8)
            params.instance_rounding_policies = instance_rounding.ctypes.data
        self._check(self._lib.aclhip_decompress_tracks_host(self._handle, clips.ctypes.data, sample_times.ctypes.data, n, ctypes.byref(params), count, out.ctypes.data, num_tracks * 48))
        return out

    def decompress_track(self, clips, sample_times, track_indices, params=None, out=None, default_values=None, track_rounding=None, instance_rounding=None):
        """Host arrays in, one qvv (12 floats) per instance out."""
        clips = np.ascontiguousarray(clips, dtype=np.uint32)
        sample_times = np.ascontiguousarray(sample_times, dtype=np.float32)
        track_indices = np.ascontiguousarray(track_indices, dtype=np.uint32)
        n = clips.size
        if out is None:
            out = np.zeros((n, 12), dtype=np.float32)
        params = params if params is not None else default_params()
        count = 0
        if default_values is not None:
            default_values = np.ascontiguousarray(default_values, dtype=np.float32)
            params.default_values = default_values.ctypes.data
            count = default_values.size // 12
        if track_rounding is not None:
            track_rounding = np.ascontiguousarray(track_rounding, dtype=np.uint8)
            params.track_rounding_policies = track_rounding.ctypes.data
        if instance_rounding is not None:
            instance_rounding = np.ascontiguousarray(instance_rounding, dtype=np.uint8)
            params.instance_rounding_policies = instance_rounding.ctypes.data
        self._check(self._lib.aclhip_decompress_track_host(self._handle, clips.ctypes.data, sample_times.ctypes.data, track_indices.ctypes.data, n, ctypes.byref(params), count, out.ctypes.data))
        return out

    # ---- scalar track lists (float1f .. vector4f) ----
    def decompress_scalar_tracks_batch(self, clips_ptr, times_ptr, num_instances, values_ptr, stride_bytes, params=None, stream=None):
        """seek + decompress_tracks of scalar track lists. All pointers are device addresses (ints)."""
        params = params if params is not None else default_params()
        self._check(self._lib.aclhip_decompress_scalar_tracks_batch(self._handle, clips_ptr, times_ptr, num_instances, ctypes.byref(params), values_ptr, stride_bytes, stream))

    def decompress_scalar_track_batch(self, clips_ptr, times_ptr, tracks_ptr, num_instances, values_ptr, stride_bytes, params=None, stream=None):
        params = params if params is not None else default_params()
        self._check(self._lib.aclhip_decompress_scalar_track_batch(self._handle, clips_ptr, times_ptr, tracks_ptr, num_instances, ctypes.byref(params), values_ptr, stride_bytes, stream))

    def _scalar_shape(self, clips):
        infos = [self.clip_info(int(c)) for c in np.unique(clips)]
        return max((i.num_tracks for i in infos), default=0), max((i.num_components for i in infos), default=1)

    def decompress_scalar_tracks(self, clips, sample_times, params=None, out=None, track_rounding=None, instance_rounding=None):
        """Host arrays in, host values out: float32 [n, max num_tracks, max num_components] (rows of clips with fewer components are packed tighter)."""
        clips = np.ascontiguousarray(clips, dtype=np.uint32)
        sample_times = np.ascontiguousarray(sample_times, dtype=np.float32)
        if out is None:
            num_tracks, num_components = self._scalar_shape(clips)
            out = np.zeros((clips.size, num_tracks, num_components), dtype=np.float32)
        params = params if params is not None else default_params()
        if track_rounding is not None:
            track_rounding = np.ascontiguousarray(track_rounding, dtype=np.uint8)
            params.track_rounding_policies = track_rounding.ctypes.data
        if instance_rounding is not None:
            instance_rounding = np.ascontiguousarray(instance_rounding, dtype=np.uint8)
            params.instance_rounding_policies = instance_rounding.ctypes.data
        stride = out.strides[0] if clips.size else 4
        self._check(self._lib.aclhip_decompress_scalar_tracks_host(self._handle, clips.ctypes.data, sample_times.ctypes.data, clips.size, ctypes.byref(params), out.ctypes.data, max(stride, 4)))
        return out

    def decompress_scalar_track(self, clips, sample_times, track_indices, params=None, out=None, track_rounding=None):
        clips = np.ascontiguousarray(clips, dtype=np.uint32)
        sample_times = np.ascontiguousarray(sample_times, dtype=np.float32)
        track_indices = np.ascontiguousarray(track_indices, dtype=np.uint32)
        if out is None:
            out = np.zeros((clips.size, self._scalar_shape(clips)[1]), dtype=np.float32)
        params = params if params is not None else default_params()
        if track_rounding is not None:
            track_rounding = np.ascontiguousarray(track_rounding, dtype=np.uint8)
            params.track_rounding_policies = track_rounding.ctypes.data
        self._check(self._lib.aclhip_decompress_scalar_track_host(self._handle, clips.ctypes.data, sample_times.ctypes.data, track_indices.ctypes.data, clips.size,
                                                                 ctypes.byref(params), out.ctypes.data, max(out.strides[0], 4) if clips.size else 4))
        return out

    def decompress_all_samples(self, clip, scratch_ptr, out_ptr, stride_bytes, params=None, stream=None):
        """convert_track_list's sampling loop: every sample of `clip`, nearest rounding. Device pointers; scratch = 8 * num_samples bytes."""
        self._check(self._lib.aclhip_decompress_all_samples(self._handle, clip, ctypes.byref(params) if params is not None else None, scratch_ptr, out_ptr, stride_bytes, stream))

    def all_gather_poses(self, rccl_comm, shard_ptr, all_ptr, shard_bytes, stream=None):
        """One RCCL all-gather of pose shards (rank order); `rccl_comm` is an ncclComm_t handle (int / c_void_p)."""
        self._check(self._lib.aclhip_all_gather_poses(self._handle, rccl_comm, shard_ptr, all_ptr, shard_bytes, stream))

    # ---- peer gather (aclhip_peer_*): one GPU collects every rank's shard over xGMI ----
    def peer_export_buffer(self, buffer_ptr):
        handle = (ctypes.c_uint8 * PEER_HANDLE_BYTES)()
        self._check(self._lib.aclhip_peer_export_buffer(self._handle, buffer_ptr, handle))
        return bytes(handle)

    def peer_open_buffer(self, handle):
        raw = (ctypes.c_uint8 * PEER_HANDLE_BYTES)(*handle)
        pointer = ctypes.c_void_p()
        self._check(self._lib.aclhip_peer_open_buffer(self._handle, raw, ctypes.byref(pointer)))
        return pointer.value

    def peer_close_buffer(self, buffer_ptr):
        self._check(self._lib.aclhip_peer_close_buffer(self._handle, buffer_ptr))

    def push_poses_to_peer(self, peer_ptr, offset_bytes, shard_ptr, shard_bytes, stream=None):
        self._check(self._lib.aclhip_push_poses_to_peer(self._handle, peer_ptr, offset_bytes, shard_ptr, shard_bytes, stream))

    def negative_scale_count(self):
        count = ctypes.c_uint64(0)
        self._check(self._lib.aclhip_get_negative_scale_count(self._handle, ctypes.byref(count)))
        return count.value

    def lifetime_stats(self):
        """aclhip_get_lifetime_stats as a dict"""
        values = (ctypes.c_uint64 * 8)()
        self._check(self._lib.aclhip_get_lifetime_stats(self._handle, values))
        names = ("registered", "unregistered", "recycled", "pending", "table_capacity", "table_is_virtual", "table_address", "launch_streams")
        return dict(zip(names, (int(v) for v in values)))

    def rejected_instance_count(self):
        count = ctypes.c_uint64(0)
        self._check(self._lib.aclhip_get_rejected_instance_count(self._handle, ctypes.byref(count)))
        return count.value

    def measure_write_bandwidth(self, buffer_ptr, size_bytes, repeats=20, stream=None):
        """GB/s of a plain 16 byte per lane store stream into the given device buffer."""
        gbps = ctypes.c_float(0.0)
        self._check(self._lib.aclhip_measure_write_bandwidth(self._handle, buffer_ptr, size_bytes, repeats, stream, ctypes.byref(gbps)))
        return gbps.value

    def measure_pose_store_bandwidth(self, poses_ptr, pose_stride_bytes, num_instances, num_tracks, repeats=20, stream=None):
        """(GB/s, waves per CU) of the pose batch's own store stream alone, best of 32 / 16 / 12 / 8 resident waves per CU"""
        gbps, waves = ctypes.c_float(0.0), ctypes.c_uint32(0)
        self._check(self._lib.aclhip_measure_pose_store_bandwidth(self._handle, ctypes.c_void_p(poses_ptr), ctypes.c_uint64(pose_stride_bytes), ctypes.c_uint32(num_instances),
                                                                  ctypes.c_uint32(num_tracks), ctypes.c_uint32(repeats), ctypes.c_void_p(stream), ctypes.byref(gbps), ctypes.byref(waves)))
        return gbps.value, waves.value

    def tracks_kernel_name(self, params=None, pose_stride_bytes=None, output=None):
        """Name of the kernel a pose launch takes: for rows as wide as the largest registered clip, or (pose_stride_bytes given) for the
        launch aclhip_decompress_tracks_batch_out makes with `output` and that stride."""
        params = params if params is not None else default_params()
        name = ctypes.create_string_buffer(128)
        if pose_stride_bytes is None and output is None:
            self._check(self._lib.aclhip_describe_tracks_kernel(self._handle, ctypes.byref(params), name, 128))
        else:
            stride = int(pose_stride_bytes) if pose_stride_bytes is not None else 0xFFFFFFFFFFFFFFFF
            self._check(self._lib.aclhip_describe_tracks_launch(self._handle, ctypes.byref(params), ctypes.byref(output) if output is not None else None, stride, name, 128, None))
        return name.value.decode()

    def batch_algorithmic_bytes(self, clips):
        clips = np.ascontiguousarray(clips, dtype=np.uint32)
        written, read = ctypes.c_uint64(0), ctypes.c_uint64(0)
        self._check(self._lib.aclhip_batch_algorithmic_bytes(self._handle, clips.ctypes.data, clips.size, ctypes.byref(written), ctypes.byref(read)))
        return written.value, read.value
