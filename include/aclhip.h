/* aclhip.h -- C ABI of the MI355X-native batched ACL clip decompressor (libaclhip.so).
 *
 * This is the drop-in boundary for the reference's decompression path. The reference has no ABI: its
 * boundary is the inlined C++ template surface of acl::decompression_context<Settings>
 * (/root/reference/includes/acl/decompression/decompress.h:76-201). Each entry point below names the
 * reference interface it replaces; acl_amd/csrc/aclhip.hpp rebuilds the reference's C++ surface on top
 * of these calls, and INTEGRATION.md shows the binding a maintainer of the reference would add.
 *
 * Conventions
 *   - plain C, no HIP or torch types: device pointers are `void*`/typed pointers into HBM of the context's
 *     device, streams are passed as `void*` (a hipStream_t; NULL = the default stream);
 *   - every function returns an aclhip_status and never throws; decompress calls are asynchronous and
 *     stream ordered (the reference's calls are synchronous CPU code);
 *   - a context owns device copies of registered clips; the caller owns instance lists and pose buffers;
 *   - a pose is `num_tracks` records of 48 bytes, the reference's rtm::qvvf:
 *     rotation xyzw | translation xyz, 0 | scale xyz, 0   (core/impl/debug_track_writer.h:61-62,172-192).
 */
#ifndef ACLHIP_H
#define ACLHIP_H

#include <stdint.h>

#if defined(__cplusplus)
extern "C" {
#endif

#define ACLHIP_VERSION_MAJOR 0
#define ACLHIP_VERSION_MINOR 1

typedef enum aclhip_status
{
	ACLHIP_OK = 0,
	ACLHIP_ERROR_INVALID_ARGUMENT = 1,
	ACLHIP_ERROR_INVALID_CLIP = 2,			/* compressed_tracks::is_valid() would fail (core/impl/compressed_tracks.impl.h:278-301) */
	ACLHIP_ERROR_UNSUPPORTED_FORMAT = 3,	/* an unknown track type or rotation format; tables beyond the limits registration states */
	ACLHIP_ERROR_UNKNOWN_CLIP = 4,
	ACLHIP_ERROR_OUT_OF_MEMORY = 5,
	ACLHIP_ERROR_DEVICE = 6,				/* a HIP call failed, see aclhip_last_error_message */
	ACLHIP_ERROR_NO_DEVICE = 7,
	ACLHIP_ERROR_UNKNOWN_DATABASE = 8,
	ACLHIP_ERROR_NOT_IN_DATABASE = 9,
	ACLHIP_ERROR_NO_METADATA = 10			/* the blob does not carry the optional metadata that was asked for */
} aclhip_status;

/* acl::sample_rounding_policy (core/sample_rounding_policy.h) */
typedef enum aclhip_rounding_policy
{
	ACLHIP_ROUND_NONE = 0,
	ACLHIP_ROUND_FLOOR = 1,
	ACLHIP_ROUND_CEIL = 2,
	ACLHIP_ROUND_NEAREST = 3,
	ACLHIP_ROUND_PER_TRACK = 4
} aclhip_rounding_policy;

/* acl::sample_looping_policy (core/sample_looping_policy.h) */
typedef enum aclhip_looping_policy
{
	ACLHIP_LOOP_CLAMP = 0,
	ACLHIP_LOOP_WRAP = 1,
	ACLHIP_LOOP_AS_COMPRESSED = 2
} aclhip_looping_policy;

/* acl::rotation_normalization_policy_t (decompression/decompression_settings.h:50-62) */
typedef enum aclhip_normalization_policy
{
	ACLHIP_NORMALIZE_NEVER = 0,
	ACLHIP_NORMALIZE_LERP_ONLY = 1,			/* default_transform_decompression_settings */
	ACLHIP_NORMALIZE_ALWAYS = 2
} aclhip_normalization_policy;

/* acl::default_sub_track_mode (core/track_writer.h:49-74) */
typedef enum aclhip_default_mode
{
	ACLHIP_DEFAULT_SKIPPED = 0,				/* default sub-tracks are not written, the caller pre-filled the pose buffer */
	ACLHIP_DEFAULT_CONSTANT = 1,			/* one value for every default sub-track (identity / 0 / 1 when no value is given) */
	ACLHIP_DEFAULT_VARIABLE = 2,			/* per track value, e.g. the bind pose */
	ACLHIP_DEFAULT_LEGACY = 3,				/* scale only: the clip's default scale bit (ACL 2.0 behaviour) */
	ACLHIP_DEFAULT_BIND_POSE = 4			/* not in the reference: variable, with every clip's OWN table -- track_desc_transformf::default_value of each track,
											 * read from the blob's optional track descriptions at registration (compressed_tracks::get_track_description,
											 * core/impl/compressed_tracks.impl.h:214-275); the identity for clips that carry none. What a caller of the
											 * reference does by hand: get_track_description() per track into a debug_track_writer_variable_defaults */
} aclhip_default_mode;

typedef struct aclhip_context aclhip_context;
typedef uint32_t aclhip_clip;				/* handle returned by aclhip_register_clip */
typedef uint32_t aclhip_database;			/* handle returned by aclhip_register_database */

#define ACLHIP_INVALID_HANDLE 0xFFFFFFFFu

/* What decompression_settings + track_writer select at compile time in the reference
 * (decompression/decompression_settings.h:74-166, core/track_writer.h:82-216), as a run time struct.
 * aclhip_default_params() fills in default_transform_decompression_settings + the track_writer defaults. */
typedef struct aclhip_decompress_params
{
	uint8_t rounding_policy;				/* aclhip_rounding_policy given to seek() for every instance (unless per-instance policies are supplied) */
	uint8_t looping_policy;					/* aclhip_looping_policy, decompression_context::set_looping_policy() */
	uint8_t normalization;					/* aclhip_normalization_policy, get_rotation_normalization_policy() */
	uint8_t per_track_rounding;				/* is_per_track_rounding_supported() */
	uint8_t default_rotation_mode;			/* aclhip_default_mode, track_writer::get_default_rotation_mode() */
	uint8_t default_translation_mode;
	uint8_t default_scale_mode;
	uint8_t reserved0;
	const float* default_values;			/* DEVICE pointer or NULL. CONSTANT: 12 floats; VARIABLE: max num_tracks * 12 floats (qvv per track) */
	const uint8_t* track_rounding_policies;	/* DEVICE pointer or NULL: track_writer::get_rounding_policy() per track, used when seeking with PER_TRACK */
	const uint8_t* instance_rounding_policies;	/* DEVICE pointer or NULL: one aclhip_rounding_policy per instance, overrides rounding_policy */
	const uint8_t* instance_looping_policies;	/* DEVICE pointer or NULL: one aclhip_looping_policy per instance, overrides looping_policy --
												 * decompression_context::set_looping_policy() belongs to ONE context = one instance (decompress.h:149) */
	/* track_writer::get_rounding_policy(policy, track_index) (core/track_writer.h:97) belongs to the writer of ONE pose as well: M tables
	 * of track_rounding_stride bytes each (a table = what track_rounding_policies is: one aclhip_rounding_policy per track), instance i
	 * seeks and decodes with table instance_rounding_tables[i] (< M, the caller's table: not checked). Overrides track_rounding_policies.
	 * Both or neither; per instance arrays are indexed by the caller's instance index (aclhip_output_desc). */
	const uint8_t* track_rounding_table;		/* DEVICE pointer or NULL */
	const uint8_t* instance_rounding_tables;	/* DEVICE pointer or NULL: one table index per instance */
	uint32_t track_rounding_stride;				/* bytes from one table to the next (>= tracks of the largest clip of the batch) */
	uint32_t flags;								/* ACLHIP_DECODE_* (reserved1 until ABI 6: aclhip_default_params has always zeroed it) */
} aclhip_decompress_params;

/* aclhip_decompress_params::flags.
 * ACLHIP_DECODE_FAST: opt in, per launch. The default kernels follow the reference's x86 arithmetic one IEEE operation at a time and are bit
 * exact with it (math/quatf.h:135-211); north_star's bar is 1e-5. With this flag an animated ROTATION is computed with the hardware's 1 ulp
 * square root / reciprocal square root and fused multiply-adds: every rotation component stays within 2e-6 of the default kernels'
 * (tests/test_gpu_fast_decode.py asserts it over every instance of the BASELINE.json batches and the corpus); the x, y, z of every sample
 * (the range expansions are never fused), constant and default sub-tracks, translations and scales are bit identical. Taken by the plain
 * decode (aclhip_decompress_tracks_batch / _rows / _list with the QVV48 layout and the track_writer defaults); launches with other settings
 * or an output descriptor keep the exact kernels, as do per track rounding policies and aclhip_decompress_track_batch (its variant never
 * measured faster than the exact kernel and was removed: the flag is accepted there and changes nothing). What it removes is vector
 * instructions of poses of several windows (the 300-bone rig: 4 % of them) -- worth 2 % of the launch while both kernels ran at 7 waves
 * per SIMD and nothing since the exact kernel fits 8 (190.7 against 190.8 us); a one-window batch sits on its write stream either way. */
#define ACLHIP_DECODE_FAST 1u

/* Where a decoded pose goes and what of it: the run time form of the OUTPUT side of the track_writer protocol
 * (core/track_writer.h:161-216). A writer decides per sub-track kind whether it wants it at all -- skip_all_rotations /
 * skip_all_translations / skip_all_scales (:181-183) -- and the writer's own pose type decides how wide a bone is: the
 * reference's debug writer stores rtm::qvvf (48 bytes, core/impl/debug_track_writer.h:61-62), its benchmark counts
 * sizeof(rtm::quatf) + 2 * sizeof(rtm::float3f) = 40 bytes per bone (tools/acl_decompressor/sources/benchmark.cpp:146), engines
 * that ignore scale keep 32. The decode sits on the HBM write roofline, so bytes per pose are poses per second. */
typedef enum aclhip_pose_layout
{
	ACLHIP_LAYOUT_QVV48 = 0,				/* per track: rotation xyzw | translation xyz 0 | scale xyz 0 (rtm::qvvf), the default */
	ACLHIP_LAYOUT_QVV40 = 1,				/* per track: rotation xyzw | translation xyz | scale xyz, 10 packed floats */
	ACLHIP_LAYOUT_QV32 = 2					/* per track: rotation xyzw | translation xyz 0; scales are not written (implies skip_scales) */
} aclhip_pose_layout;

typedef struct aclhip_output_desc
{
	uint32_t layout;						/* aclhip_pose_layout */
	uint8_t skip_rotations;					/* track_writer::skip_all_rotations(): no rotation is written, its bytes in the pose buffer are left untouched */
	uint8_t skip_translations;				/* track_writer::skip_all_translations() */
	uint8_t skip_scales;					/* track_writer::skip_all_scales() */
	uint8_t reserved0;
	const uint32_t* rows;					/* DEVICE pointer or NULL: pose of instance i goes to row rows[i] (distinct) instead of row i */
	const uint8_t* skip_tracks;				/* DEVICE pointer or NULL: track_writer::skip_track_rotation / _translation / _scale(track_index)
											 * (core/track_writer.h:189-191) for the whole launch: one byte per track (as many as the largest clip of
											 * the batch has tracks), bit 0 / 1 / 2 set = the track's rotation / translation / scale is skipped -- not
											 * written, its bytes in the pose buffer are left untouched (an LOD that drops finger bones) */
	/* Per INSTANCE writer decisions (ABI 5). In the reference the track_writer belongs to ONE decompress_tracks call, i.e. to one pose:
	 * skip_track_rotation / _translation / _scale(track_index) (core/track_writer.h:189-191) are per character. A crowd with per
	 * character LODs is ONE launch here:
	 *   mask_table + instance_masks   M skip masks of mask_stride bytes each (a mask = what skip_tracks is: one byte per track, bit 0 / 1 / 2 =
	 *                                 rotation / translation / scale skipped); instance i uses mask instance_masks[i] (< M, not checked: the
	 *                                 caller's table). Overrides skip_tracks.
	 *   instance_track_counts         instance i stores only its first instance_track_counts[i] tracks (the LOD most engines use: bones are
	 *                                 ordered by importance); nothing beyond is decoded, DMA'd or written -- on a kernel bound by its
	 *                                 writes, bytes not written are poses per second. A count above the clip's track count changes nothing.
	 * Every per instance array of this struct and of aclhip_decompress_params is indexed by the CALLER's instance index (instance lists:
	 * the index in the list handed to aclhip_instance_list_set_clips, whatever order the library decodes in). */
	const uint8_t* mask_table;				/* DEVICE pointer or NULL */
	const uint8_t* instance_masks;			/* DEVICE pointer or NULL: one mask index per instance; needs mask_table */
	const uint32_t* instance_track_counts;	/* DEVICE pointer or NULL */
	uint32_t mask_stride;					/* bytes from one mask of mask_table to the next (>= tracks of the largest clip of the batch) */
	uint32_t reserved1;
} aclhip_output_desc;

typedef struct aclhip_clip_info
{
	uint32_t num_tracks;
	uint32_t num_samples;
	float sample_rate;
	float duration;							/* compressed_tracks::get_finite_duration(as_compressed) */
	uint32_t num_segments;
	uint32_t has_scale;
	uint32_t looping_policy;				/* compressed_tracks::get_looping_policy() */
	uint32_t compressed_size;				/* compressed_tracks::get_size() */
	uint32_t hash;							/* compressed_tracks::get_hash() */
	uint32_t num_animated_sub_tracks;		/* rotations + translations + scales */
	uint32_t has_database;
	uint32_t has_stripped_keyframes;
	uint32_t track_type;					/* acl::track_type8 (core/track_types.h:51-68): 12 qvvf; 0..4 float1f, float2f, float3f, float4f, vector4f */
	uint32_t num_components;				/* floats per sample of a track: 12 for qvvf (one qvv record), 1..4 for scalar track lists */
} aclhip_clip_info;

/* ---- library / context ------------------------------------------------------------------------ */

const char* aclhip_status_string(aclhip_status status);

/* Message of the last failing call made on the CALLING thread (empty string when none); `context` is not used to find it. */
const char* aclhip_last_error_message(const aclhip_context* context);

/* The layouts of the structs in this header as a number: bumped whenever one of them changes (3: aclhip_output_desc::skip_tracks;
 * 4: aclhip_pose_consumers::num_blend_clips, flags, blend_clips, blend_sample_times, blend_weights;
 * 5: aclhip_decompress_params::instance_looping_policies, track_rounding_table, instance_rounding_tables, track_rounding_stride, aclhip_output_desc::mask_table, instance_masks, instance_track_counts, mask_stride;
 * 6: ACLHIP_DEFAULT_BIND_POSE, aclhip_clip_metadata_info; ACLHIP_ERROR_UNSUPPORTED_FORMAT no longer covers the full-precision formats).
 * A caller compiled against another header would hand over structs of another shape; aclhip_abi_version() says what the LIBRARY was
 * built with, and the C++ mirror (aclhip.hpp) refuses to create a context when the two differ. */
#define ACLHIP_ABI_VERSION 6u
uint32_t aclhip_abi_version(void);

/* Creates a context bound to HIP device `device_index` (replaces nothing in the reference: contexts there are
 * 128 byte stack objects, decompression/impl/decompression_context.transform.h:53-116). */
aclhip_status aclhip_create(int device_index, aclhip_context** out_context);
void aclhip_destroy(aclhip_context* context);

void aclhip_default_params(aclhip_decompress_params* out_params);

/* ---- clips ------------------------------------------------------------------------------------ */

/* Replaces decompression_context::initialize(const compressed_tracks&) (decompress.h:103; impl/decompress.impl.h:66-83;
 * initialize_v0 impl/decompression.transform.h:84-132): validates the blob like compressed_tracks::is_valid(check_hash)
 * and copies it, unchanged and 16 byte aligned with tail padding, into HBM together with derived lookup tables.
 * `compressed_tracks` is a HOST pointer to `size` bytes; the caller may free it as soon as the call returns. */
aclhip_status aclhip_register_clip(aclhip_context* context, const void* compressed_tracks, uint64_t size, int check_hash, aclhip_clip* out_clip);

/* Replaces decompression_context::reset() / the end of the blob's lifetime. Like every call that registers, replaces or retires
 * something (clips, hierarchies, databases) it never synchronizes the device and nobody waits: decodes that were ENQUEUED before the
 * call, on any stream this context has launched on, still decode the clip (its record in the device table is cleared behind them, on
 * a stream of the context's own); launches that execute later refuse the handle (counted, poses untouched); the clip's memory and its
 * handle are recycled when both have happened. The caller's side of the contract is the reference's: do not enqueue a decode of a
 * clip after its unregistration. */
aclhip_status aclhip_unregister_clip(aclhip_context* context, aclhip_clip clip);

/* The context remembers every stream it has launched on (retired clips, hierarchies and databases wait for the work enqueued on
 * them). Call this BEFORE destroying a stream the context has launched on: it waits for that stream's work (hipStreamSynchronize)
 * and forgets the stream. A destroyed stream the context still remembers is detected when its next event cannot be recorded, but
 * using a stale handle is undefined behaviour by HIP's rules -- this call is the defined way. */
aclhip_status aclhip_forget_stream(aclhip_context* context, void* stream);

/* Counters of the stream ordered lifetime management, for tests and tools: out_stats[0] clips registered, [1] clips unregistered,
 * [2] retired items whose memory has been recycled, [3] retired items still waiting for work in flight, [4] capacity of the clip
 * table in records, [5] 1 when the table grows inside a reserved address range (it never moves either way), [6] its device
 * address, [7] streams the context has launched on. out_stats holds 8 values. */
aclhip_status aclhip_get_lifetime_stats(aclhip_context* context, uint64_t* out_stats);

aclhip_status aclhip_get_clip_info(const aclhip_context* context, aclhip_clip clip, aclhip_clip_info* out_info);

/* The optional metadata a blob may carry behind its compressed data (compression_metadata_settings, compression_settings.h:84-120): read and
 * bounds checked once, at registration. Replaces compressed_tracks::get_parent_track_index / get_track_description
 * (core/impl/compressed_tracks.impl.h:175-275) for registered clips. */
typedef struct aclhip_clip_metadata_info
{
	uint32_t has_metadata;					/* tracks_header::get_has_metadata() */
	uint32_t has_parent_track_indices;		/* ... and the section is stored (and lies inside the blob) */
	uint32_t has_track_descriptions;
	uint32_t has_track_names;
	uint32_t has_track_list_name;
	uint32_t has_contributing_error;
} aclhip_clip_metadata_info;
aclhip_status aclhip_get_clip_metadata_info(const aclhip_context* context, aclhip_clip clip, aclhip_clip_metadata_info* out_info);

/* Host only (no context, no device), on a blob that need not be registered -- where the reference's accessors live: which optional sections
 * the blob stores and, for the arrays that are not null (capacity = the tracks they hold), get_parent_track_index / get_track_description of
 * every track as the two calls below return them. Sections that are not stored leave their arrays untouched. */
aclhip_status aclhip_read_clip_metadata(const void* compressed_tracks, uint64_t size, aclhip_clip_metadata_info* out_info, uint32_t* out_parent_indices,
	float* out_default_values, float* out_precisions, float* out_shell_distances, uint32_t capacity);

/* compressed_tracks::get_parent_track_index for every track (ACLHIP_NO_PARENT = k_invalid_track_index); `capacity` entries are available at
 * out_parent_indices and must cover the clip's tracks. ACLHIP_ERROR_NO_METADATA when the blob does not store them. */
aclhip_status aclhip_get_clip_parent_indices(const aclhip_context* context, aclhip_clip clip, uint32_t* out_parent_indices, uint32_t capacity);

/* compressed_tracks::get_track_description(track, track_desc_transformf&) for every track: default_value as 12 floats per track (rotation
 * xyzw | translation xyz 0 | scale xyz 0: a row of aclhip_decompress_params::default_values), and -- both optional -- precision and
 * shell_distance. `capacity` = tracks the arrays hold. ACLHIP_ERROR_NO_METADATA when the blob does not store descriptions. */
aclhip_status aclhip_get_clip_track_descriptions(const aclhip_context* context, aclhip_clip clip, float* out_default_values, float* out_precisions, float* out_shell_distances, uint32_t capacity);

/* Replaces compressed_tracks::is_valid(check_hash) (core/impl/compressed_tracks.impl.h:278-301) as a host only call (no context,
 * no device): everything aclhip_register_clip checks before it uploads -- tag, version, hash, every header offset, sub-track
 * classes against counts, bit widths against the per segment pose size, stored keyframes inside the buffer. `out_message`
 * (optional) receives the reason, like error_result::c_str(). The reference only checks alignment, tag, version and hash; the
 * rest is here because the device reads through these offsets -- and, since round 5, because a buffer that is accepted has to decode to
 * the SAME poses here and in the reference: also refused are blobs the reference's decoder and these kernels would read differently
 * (a segment whose sample range claims keyframes that overlap the next segment's data; a stripped segment that does not keep its first
 * and last sample; segment start indices without their 0xFFFFFFFF end or that the reference's guess-and-scan lookup would resolve to
 * other segments; per segment rotation / translation bit sizes and constant sample counts that disagree with the sub-track types;
 * section offsets that are not 4 byte aligned). Everything the reference's compressor writes passes. */
aclhip_status aclhip_check_clip(const void* compressed_tracks, uint64_t size, int check_hash, char* out_message, uint32_t capacity);

/* Host only, like aclhip_check_clip: what registration derives about the VALUES a (valid) clip can decode to, which decides the kernel
 * variants its instances run -- for tools ("do my clips take the short arithmetic?") and tests. Bits of *out_facts:
 *   ACLHIP_CLIP_FACT_SHORT_EXACT_MATH  no quantized rotation sample can hand the kernels a square root argument in (0, 2^-96) or a
 *                                      norm outside [2^-126, 2^126]: the short correctly rounded square root / reciprocal run
 *                                      (same bits, fewer instructions: DESIGN.md 4.1);
 *   ACLHIP_CLIP_FACT_RAW_ROTATIONS     some rotation sub-track is stored raw (fp32) in some segment;
 *   ACLHIP_CLIP_FACT_NEGATIVE_SCALE    some scale may decode to a negative component: the pose consumers compile rtm::qvv_mul's
 *                                      matrix route in while such a clip is registered.
 * Scalar track lists: 0. No reference counterpart (the reference has one code path). */
#define ACLHIP_CLIP_FACT_SHORT_EXACT_MATH 1u
#define ACLHIP_CLIP_FACT_RAW_ROTATIONS 2u
#define ACLHIP_CLIP_FACT_NEGATIVE_SCALE 4u
aclhip_status aclhip_analyze_clip(const void* compressed_tracks, uint64_t size, int check_hash, uint32_t* out_facts);

/* Replaces decompression_context::is_bound_to(const compressed_tracks&) (decompress.h:138): true when `clip`
 * was registered from a blob with the same hash and size. */
aclhip_status aclhip_clip_matches(const aclhip_context* context, aclhip_clip clip, const void* compressed_tracks, int* out_matches);

/* ---- databases (streamed keyframe tiers) ------------------------------------------------------- */

typedef struct aclhip_database_info
{
	uint32_t num_clips;
	uint32_t num_segments;
	uint32_t max_chunk_size;
	uint32_t num_chunks[2];					/* [0] medium importance tier, [1] low importance tier */
	uint32_t num_loaded_chunks[2];
	uint32_t bulk_data_size[2];
} aclhip_database_info;

/* Replaces database_context::initialize(allocator, database, medium_streamer, low_streamer)
 * (decompression/database/database.h:116; impl/database.impl.h): registers a compressed_database (HOST pointer, `size` bytes).
 * Bulk data of the two tiers is given separately (split_database_bulk_data) or may be null when it is inline in the database.
 * A bulk data pointer given separately must address compressed_database::get_bulk_data_size(tier) bytes (database_header::
 * bulk_data_size, aclhip_database_info::bulk_data_size after aclhip_check_database): like the reference's streamers, the call takes
 * no size for it and trusts the header the caller paired it with.
 * The bulk data is copied once into PINNED host memory -- the streamer's backing store -- and HBM buffers of the same size are
 * reserved; nothing is resident on the GPU until aclhip_database_stream_in. The runtime tier metadata the decoder reads
 * (database_runtime_segment_header::tier_metadata, core/impl/compressed_headers.h:404-422) lives in HBM. */
aclhip_status aclhip_register_database(aclhip_context* context, const void* compressed_database, uint64_t size,
	const void* bulk_data_medium, const void* bulk_data_low, int check_hash, aclhip_database* out_database);
aclhip_status aclhip_unregister_database(aclhip_context* context, aclhip_database database);

/* Replaces compressed_database::is_valid(check_hash) (core/impl/compressed_database.impl.h:142-163) as a host only call, with the
 * checks of aclhip_register_database (chunk and segment headers inside the bulk data, bulk data hashes). */
aclhip_status aclhip_check_database(const void* compressed_database, uint64_t size, const void* bulk_data_medium, const void* bulk_data_low,
	int check_hash, char* out_message, uint32_t capacity);
aclhip_status aclhip_get_database_info(const aclhip_context* context, aclhip_database database, aclhip_database_info* out_info);

/* Replaces decompression_context::initialize(const compressed_tracks&, const database_context&) (decompress.h:108;
 * impl/decompress.impl.h:85-113): like aclhip_register_clip for a clip that database.contains(). */
aclhip_status aclhip_register_clip_with_database(aclhip_context* context, const void* compressed_tracks, uint64_t size, int check_hash,
	aclhip_database database, aclhip_clip* out_clip);

/* Replace database_context::stream_in / stream_out(tier, num_chunks) (database/database.h:160-181, impl/database.impl.h:443-640):
 * tier 1 = medium importance, 2 = lowest importance. stream_in copies the next `num_chunks` missing chunks from pinned host
 * memory to HBM with hipMemcpyAsync on `stream` and then publishes their segments' tier metadata (stream ordered: decodes enqueued
 * later on the same stream see the new keyframes); stream_out retires the metadata of the first `num_chunks` resident chunks.
 * `out_num_chunks` (optional) receives how many chunks were actually moved (0 = done, like database_stream_request_result::done).
 * A request for 0 chunks follows the reference's arithmetic to the letter (database.impl.h:490-497,571-578: `first + 0 - 1` wraps
 * when the first candidate is chunk 0 and the WHOLE tier moves; with any other first candidate nothing does). */
aclhip_status aclhip_database_stream_in(aclhip_context* context, aclhip_database database, uint32_t tier, uint32_t num_chunks, void* stream, uint32_t* out_num_chunks);
aclhip_status aclhip_database_stream_out(aclhip_context* context, aclhip_database database, uint32_t tier, uint32_t num_chunks, void* stream, uint32_t* out_num_chunks);

/* Databases whose bulk data is served by the CALLER's streamers (acl::database_streamer objects, decompression/database/
 * database_streamer.h:95-175): only the compressed_database itself is known at registration. A stream-in request then hands over the
 * tier's bulk data as the streamer holds it (database_streamer::get_bulk_data(tier), HOST pointer; valid for the chunks the request
 * selects -- the same chunks the reference's database_context selects for the same request, database.impl.h:478-497): chunks seen for
 * the first time are validated and turned into tier metadata like aclhip_register_database does up front, their bytes are copied and
 * travel to HBM on `stream`. aclhip_database_stream_out and everything else work as for any database.
 * acl_gpu::database_context (acl_amd/csrc/acl_gpu_adapter.h) drives the caller's streamers through the reference's own
 * database_context and mirrors every completed request with these calls. */
aclhip_status aclhip_register_database_streamed(aclhip_context* context, const void* compressed_database, uint64_t size, int check_hash, aclhip_database* out_database);
aclhip_status aclhip_database_stream_in_from(aclhip_context* context, aclhip_database database, uint32_t tier, uint32_t num_chunks, const void* tier_bulk_data,
	void* stream, uint32_t* out_num_chunks);

/* Host only (no GPU work): strip_database_quality_tier (compression/compress.h:124, impl/compress.database.impl.h:1388-1525) --
 * the compressed_database without its medium (tier 1) or low (tier 2) importance tier, byte for byte what the reference builds
 * (it reserves room for the remaining tier's bulk data and sets its offset whether or not the bulk data is inline, and copies
 * only inline bulk data; so does this). The input must pass is_valid(true). Call with out_database NULL / capacity 0 for the
 * size. ACLHIP_ERROR_INVALID_ARGUMENT for the high importance tier (0) and for an empty tier, like the reference's errors. */
aclhip_status aclhip_strip_database_tier(const void* compressed_database, uint64_t size, uint32_t tier, void* out_database, uint64_t capacity, uint64_t* out_size);

/* ---- decompression ---------------------------------------------------------------------------- */

/* Replaces, for every instance i in [0, num_instances):
 *     context.seek(sample_times[i], rounding_policy);            (decompress.h:160; seek_v0 impl/decompression.transform.h:206-563)
 *     context.decompress_tracks(writer);                          (decompress.h:166; decompress_tracks_v0 :1526-1737)
 * with writer.write_rotation/translation/scale storing into
 *     (char*)poses + i * pose_stride_bytes + track_index * 48.
 * clips / sample_times / poses are DEVICE pointers; pose_stride_bytes must be a multiple of 16 and at least
 * 48 * num_tracks of the largest clip referenced. One wavefront decodes one window of 312 pose quads (104 tracks) of one instance.
 *
 * A launch is shaped by its BATCH: a row of pose_stride_bytes holds at most pose_stride_bytes / 48 tracks, so that -- or the largest
 * registered clip, whichever is smaller -- decides how many wavefronts an instance gets and how much LDS each of them; what else the
 * context holds (a 551-bone crowd leader next to the 100-bone characters of this batch) does not matter. The reference sizes its work
 * per clip (impl/decompression.transform.h:1526-1540). The kernels check every clip they meet against the launch and REFUSE -- count,
 * leave the pose row untouched: the reference's silent return, :1532-1537 -- an instance whose clip has more tracks than the stride
 * holds, more pose windows than the launch has wavefronts for or a window larger than the launch's LDS slots: a caller's stride that
 * is too small never writes outside its row, and a captured hipGraph replayed after a LARGER clip was registered refuses that
 * clip's instances (aclhip_get_rejected_instance_count) instead of decoding them into a launch shaped before the clip existed. Keep
 * rows as narrow as the batch needs: a stride of 14 400 bytes makes every instance a three-wavefront job. */
aclhip_status aclhip_decompress_tracks_batch(aclhip_context* context, const aclhip_clip* clips, const float* sample_times, uint32_t num_instances,
	const aclhip_decompress_params* params, void* poses, uint64_t pose_stride_bytes, void* stream);

/* Same, with the pose of instance i stored at row rows[i] of the pose buffer instead of row i (`rows`: DEVICE array of
 * num_instances distinct row indices): separates the order in which instances are decoded from where their poses go.
 * Scattered rows cost write locality: 256 clips decoded in aclhip_order_instances_for_locality order take 50 us with their
 * poses in decode order and 62 us scattered back to the original rows -- prefer numbering the rows in decode order. */
aclhip_status aclhip_decompress_tracks_batch_rows(aclhip_context* context, const aclhip_clip* clips, const float* sample_times, const uint32_t* rows,
	uint32_t num_instances, const aclhip_decompress_params* params, void* poses, uint64_t pose_stride_bytes, void* stream);

/* aclhip_decompress_tracks_batch with an output descriptor (NULL = QVV48, nothing skipped, row i): the pose of instance i starts at
 * (char*)poses + row * pose_stride_bytes and holds num_tracks records of 48 / 40 / 32 bytes in the chosen layout;
 * pose_stride_bytes must be a multiple of 16 and at least that size -- and should be a multiple of 64 (the HBM access granule): with
 * rows that start between granules every 1 KiB store of every other pose straddles them (QVV40, 100 bones: 4000 byte rows 82 us per
 * 64k poses, 4032 byte rows 44 us). Same values as the QVV48 decode, compared through the layout. */
aclhip_status aclhip_decompress_tracks_batch_out(aclhip_context* context, const aclhip_clip* clips, const float* sample_times, uint32_t num_instances,
	const aclhip_decompress_params* params, const aclhip_output_desc* output, void* poses, uint64_t pose_stride_bytes, void* stream);

/* Bytes one track takes in a pose of `layout` (48 / 40 / 32); 0 for an unknown layout. */
uint32_t aclhip_layout_bytes_per_track(uint32_t layout);

/* Host only (no GPU work): a decode order for a batch that draws on many clips -- a permutation of [0, num_instances) for the
 * instance list `clips` (HOST array) under which every clip is decoded on ONE XCD (workgroup b of a launch runs on XCD b % 8,
 * each XCD has its own L2; a clip that straddles the boundary between two XCDs' shares is decoded on both), next to its other
 * instances. Use: clips'[k] = clips[out_order[k]], sample_times'[k] = sample_times[out_order[k]], pose k of the launch belongs
 * to instance out_order[k]. 64k instances over 256 clips: 61 -> 50 us, the time of a single-clip batch; a batch of one clip is
 * unaffected. `context` tells how many wavefronts a pose of the largest registered clip takes (may be NULL: one).
 * Stable: instances of one clip keep their relative order. */
aclhip_status aclhip_order_instances_for_locality(const aclhip_context* context, const aclhip_clip* clips, uint32_t num_instances, uint32_t* out_order);

/* aclhip_order_instances_for_locality for a caller that knows the pose size instead of holding a context: windows_per_instance =
 * ceil(3 * num_tracks of the largest clip / ACLHIP_WINDOW_QUADS) wavefronts per pose (1 up to 104 tracks). Host only. */
aclhip_status aclhip_order_instances_for_pose_windows(uint32_t windows_per_instance, const aclhip_clip* clips, uint32_t num_instances, uint32_t* out_order);

/* Wavefronts per instance of the launch aclhip_decompress_tracks_batch[_out] makes for poses of `layout` in rows of `pose_stride_bytes`
 * with the clips registered now: min(largest registered clip, tracks the row holds) in pose windows of 104 tracks. Which slot of a
 * launch runs on which XCD follows from it: the value to order an instance list for (aclhip_order_instances_for_pose_windows,
 * aclhip_order_instances_device_for_windows). aclhip_order_instances_for_locality / _device assume rows as wide as the largest
 * registered clip. */
aclhip_status aclhip_pose_windows_of_launch(aclhip_context* context, uint32_t layout, uint64_t pose_stride_bytes, uint32_t* out_windows_per_instance);

/* The same order computed on the GPU for instance lists that live there (all pointers DEVICE pointers, stream ordered: ONE launch
 * on `stream` -- at most 64 workgroups that meet at barriers in global memory; three launches for registries of more than 8 192
 * clips --, scratch kept per stream by the context at a fixed size and address, no host synchronization). The one launch form needs all
 * its workgroups resident together: its grid is sized from the occupancy query to fit an idle device many times over. Should a barrier
 * not open within seconds all the same (dozens of such launches of other processes sharing the device), the launch GIVES UP without
 * placing anything -- no trap, the queue stays healthy --, the next ordering call on the stream returns ACLHIP_ERROR_DEVICE (once: the
 * order that launch was to write is invalid, order again) and the stream uses the three launch form from then on. Writes the permutation to out_order and,
 * when the pointers are not NULL, the permuted lists out_clips[k] = clips[out_order[k]], out_sample_times[k] =
 * sample_times[out_order[k]] (the arguments of the decode that follows on the same stream; rows = out_order puts the poses back
 * in the caller's rows). Which instance of a clip takes which of the clip's slots is decided by atomics: every call returns a valid
 * order, not the same one. The first call on a stream allocates that stream's scratch: make it before capturing the stream into a
 * hipGraph. A captured ordering HOLDS the scratch of the stream it was captured on: launch the graph on that stream (or at least never
 * while that stream, or another replay, orders -- two orderings that share a scratch at the same time write each other's counters; such
 * a launch places nothing out of bounds and raises the same failure as a barrier that does not open, but its order is not valid).
 * (PyTorch's CUDAGraph.replay() launches on the CURRENT stream, not on the capture stream: wrap it in `with torch.cuda.stream(s)`.)
 * An instance list usually outlives a frame (which character plays which clip changes rarely, the
 * sample times every frame): order once, keep the lists in that order. */
aclhip_status aclhip_order_instances_device(aclhip_context* context, const aclhip_clip* clips, const float* sample_times, uint32_t num_instances,
	uint32_t* out_order, aclhip_clip* out_clips, float* out_sample_times, void* stream);
/* The same for launches of `windows_per_instance` wavefronts per pose (aclhip_pose_windows_of_launch) instead of the largest registered clip's. */
aclhip_status aclhip_order_instances_device_for_windows(aclhip_context* context, uint32_t windows_per_instance, const aclhip_clip* clips, const float* sample_times,
	uint32_t num_instances, uint32_t* out_order, aclhip_clip* out_clips, float* out_sample_times, void* stream);

/* ---- persistent instance lists: the library keeps the decode order -------------------------------------------------------------
 * (No reference counterpart: the reference decodes one pose per call. SURVEY.md section 7: "sort/bucket instances by clip for L2
 * locality".) A batch that draws on hundreds of clips decodes a fifth faster in locality order (aclhip_order_instances_for_locality),
 * but ordering a list costs more than one decode of it gains. Which character plays which clip changes rarely, the sample times
 * every frame: an instance list object keeps the clip assignment of `num_instances` instances IN DECODE ORDER across frames.
 *   aclhip_instance_list_set_clips   every instance's clip (device array in the caller's instance order); orders the list on `stream`
 *   aclhip_instance_list_update      instances[k] now plays clips[k] (device arrays, `count` entries, stream ordered). The instance keeps
 *                                    its slot; once an eighth of the list has changed since it was last ordered, the next decode
 *                                    re-orders it first (one launch, on the decode's stream)
 *   aclhip_decompress_tracks_list    decodes the list: sample_times[i] is instance i's sample time (the caller's order, gathered through
 *                                    the list's order by the decode itself). Poses land in SLOT order (pose row j = instance order[j],
 *                                    aclhip_instance_list_get_order) -- 1 KiB stores to consecutive rows, what the write path likes -- or,
 *                                    with poses_in_instance_order != 0, in row i for instance i (scattered rows: measured 20 % slower).
 *                                    `output` as in aclhip_decompress_tracks_batch_out (its `rows` must be NULL), or NULL.
 *   aclhip_instance_list_attach      instead of set_clips + update: the list decodes the CALLER's own clip array (device, `num_instances`
 *                                    entries in the caller's instance order, valid and in place for as long as the list is attached to
 *                                    it). The caller's animation graph writes clip changes straight into that array -- no update launch,
 *                                    no copy: a decode reads caller_clips[order[j]] for slot j (round 4 measured the library's own update
 *                                    launch at 5.3 us per frame for 655 changed instances; the extra dependent load this form costs a
 *                                    wavefront is hidden). An instance that changed clip is still decoded correctly, just no longer next
 *                                    to its clip's other instances;
 *   aclhip_instance_list_note_changes  tells an attached list that `count` of its instances changed clip since the last call (a host
 *                                    side number, nothing is read): once an eighth of the list has changed, the next decode re-orders it
 *                                    from the caller's array as it is then.
 * A list is ordered for the shape of the launches that decode it (wavefronts per pose: aclhip_pose_windows_of_launch); a decode whose pose
 * stride gives another shape than the list was last ordered for re-orders it first.
 * All calls of one list must be made in stream order (one stream, or the caller's events between streams). WHEN a list is re-ordered
 * is decided on the host at the time of the call: a decode captured into a hipGraph replays what was decided when it was captured
 * (capture aclhip_order_instances_device + aclhip_decompress_tracks_batch instead when the order has to follow the replays' data).
 * An update must not name an instance twice: the two entries race, and the clip that plays until the next re-order need not be the one
 * that plays after it (memory safe, but undefined which). Clip handles are not validated here: the decode refuses unknown ones (counted). */
typedef uint32_t aclhip_instance_list;

aclhip_status aclhip_instance_list_create(aclhip_context* context, uint32_t num_instances, aclhip_instance_list* out_list);
aclhip_status aclhip_instance_list_destroy(aclhip_context* context, aclhip_instance_list list);
aclhip_status aclhip_instance_list_set_clips(aclhip_context* context, aclhip_instance_list list, const aclhip_clip* clips, void* stream);
aclhip_status aclhip_instance_list_update(aclhip_context* context, aclhip_instance_list list, const uint32_t* instances, const aclhip_clip* clips, uint32_t count, void* stream);
aclhip_status aclhip_instance_list_attach(aclhip_context* context, aclhip_instance_list list, const aclhip_clip* caller_clips, void* stream);
aclhip_status aclhip_instance_list_note_changes(aclhip_context* context, aclhip_instance_list list, uint32_t count);
aclhip_status aclhip_decompress_tracks_list(aclhip_context* context, aclhip_instance_list list, const float* sample_times, const aclhip_decompress_params* params,
	const aclhip_output_desc* output, int poses_in_instance_order, void* poses, uint64_t pose_stride_bytes, void* stream);
/* the list's order, slot -> instance (device pointer, num_instances entries; contents change when the list is re-ordered, stream
 * ordered with the decodes) and how often the list has been (re-)ordered so far */
aclhip_status aclhip_instance_list_get_order(aclhip_context* context, aclhip_instance_list list, const uint32_t** out_order, uint64_t* out_num_orderings);

/* Replaces seek() + decompress_track(track_indices[i], writer) (decompress.h:172; decompress_track_v0 :1753-2050):
 * one 48 byte qvv per instance at (char*)transforms + i * 48. All pointers are DEVICE pointers.
 * Any order of requests is decoded; the ORDER decides what the launch fetches: 64 consecutive requests share a wavefront, and requests
 * of many clips are best bucketed by clip -- aclhip_order_track_requests_for_locality above gives the order (4 M requests over 256
 * clips: 173 us as drawn, 77 us sorted by clip = the time of one clip, 68 - 74 us in the library's order; profiles/r06_experiments.md 5b). */
/* Host only (no GPU work): the order of a single track request list that draws on many clips -- a permutation of [0, num_requests)
 * for the request list `clips` (HOST array: the clip of every request) under which the requests are bucketed by clip (stable) and every
 * clip's requests run on ONE XCD (workgroup b of aclhip_decompress_track_batch takes requests 256 b .. 256 b + 255 and runs on XCD
 * b % 8; each XCD has its own L2). Use: clips'[k] = clips[out_order[k]], likewise sample times and track indices; transform k of the
 * launch belongs to request out_order[k]. For request lists that persist across frames (the same bones of the same characters, new
 * sample times): 4 M requests over 256 clips, 173 us as drawn, 77 us sorted by clip, 68 - 74 us in this order (HBM traffic 4.0 x the
 * algorithmic bytes as drawn, 0.98 x in this order). */
aclhip_status aclhip_order_track_requests_for_locality(const aclhip_clip* clips, uint32_t num_requests, uint32_t* out_order);

aclhip_status aclhip_decompress_track_batch(aclhip_context* context, const aclhip_clip* clips, const float* sample_times, const uint32_t* track_indices,
	uint32_t num_instances, const aclhip_decompress_params* params, void* transforms, void* stream);

/* Convenience for host callers (the C++ mirror of decompression_context uses it with a batch of one): same as the two
 * calls above but every pointer is a HOST pointer; instance lists are uploaded, poses downloaded, the call is synchronous.
 * params->default_values / track_rounding_policies / instance_rounding_policies are HOST pointers here as well;
 * `default_values_count` is the number of qvv records default_values holds (1 for CONSTANT, num_tracks for VARIABLE). */
aclhip_status aclhip_decompress_tracks_host(aclhip_context* context, const aclhip_clip* clips, const float* sample_times, uint32_t num_instances,
	const aclhip_decompress_params* params, uint32_t default_values_count, void* poses, uint64_t pose_stride_bytes);
aclhip_status aclhip_decompress_track_host(aclhip_context* context, const aclhip_clip* clips, const float* sample_times, const uint32_t* track_indices,
	uint32_t num_instances, const aclhip_decompress_params* params, uint32_t default_values_count, void* transforms);
/* aclhip_decompress_tracks_host with an output descriptor (output->rows, output->skip_tracks: HOST pointers or NULL; skip_tracks holds one
 * byte per track of the largest clip in the list). Skipped sub-tracks keep what `poses` held. */
aclhip_status aclhip_decompress_tracks_host_out(aclhip_context* context, const aclhip_clip* clips, const float* sample_times, uint32_t num_instances,
	const aclhip_decompress_params* params, uint32_t default_values_count, const aclhip_output_desc* output, void* poses, uint64_t pose_stride_bytes);

/* ---- scalar track lists (float1f / float2f / float3f / float4f / vector4f) ------------------------
 * aclhip_register_clip accepts them like transform clips (decompression_context::initialize dispatches on the track type,
 * impl/decompress.impl.h:66-83 -> initialize_v0 impl/decompression.scalar.h:100-126; databases are not supported for them).
 *
 * Replace seek() + decompress_tracks(writer) for scalar track lists (seek_v0 / decompress_tracks_v0,
 * impl/decompression.scalar.h:182-480): instance i = (clips[i], sample_times[i]); track t of instance i is written as
 * num_components floats at (char*)values + i * stride_bytes + t * num_components * 4 -- what track_writer::write_float1 /
 * write_float2 / write_float3 / write_float4 / write_vector4 (core/track_writer.h:101-158) receive. Only the rounding / looping /
 * per track rounding members of `params` apply. Transform clips in the list are rejected (and counted), as scalar clips are by the
 * transform entry points. All pointers are DEVICE pointers; asynchronous on `stream`. */
aclhip_status aclhip_decompress_scalar_tracks_batch(aclhip_context* context, const aclhip_clip* clips, const float* sample_times, uint32_t num_instances,
	const aclhip_decompress_params* params, void* values, uint64_t stride_bytes, void* stream);

/* Replaces seek() + decompress_track(track_indices[i], writer) for scalar track lists (decompress_track_v0,
 * impl/decompression.scalar.h:482-715): num_components floats per instance at (char*)values + i * stride_bytes. */
aclhip_status aclhip_decompress_scalar_track_batch(aclhip_context* context, const aclhip_clip* clips, const float* sample_times, const uint32_t* track_indices,
	uint32_t num_instances, const aclhip_decompress_params* params, void* values, uint64_t stride_bytes, void* stream);

/* The same with HOST pointers, synchronous (the C++ mirror of decompression_context uses them with a batch of one). */
aclhip_status aclhip_decompress_scalar_tracks_host(aclhip_context* context, const aclhip_clip* clips, const float* sample_times, uint32_t num_instances,
	const aclhip_decompress_params* params, void* values, uint64_t stride_bytes);
aclhip_status aclhip_decompress_scalar_track_host(aclhip_context* context, const aclhip_clip* clips, const float* sample_times, const uint32_t* track_indices,
	uint32_t num_instances, const aclhip_decompress_params* params, void* values, uint64_t stride_bytes);

/* ---- every sample of a clip --------------------------------------------------------------------- */

/* The sampling loop of convert_track_list(allocator, compressed_tracks, track_array&) (compression/convert.h:52;
 * impl/convert.impl.h:150-260): for every sample i of the clip, seek(min(float(i) / sample_rate, duration), nearest) +
 * decompress_tracks, i.e. the clip's keyframes as the decoder sees them. Row i of `out` (DEVICE, `stride_bytes` apart) receives
 * what aclhip_decompress_tracks_batch / aclhip_decompress_scalar_tracks_batch write for one instance. `scratch` is a DEVICE
 * buffer of 8 * num_samples bytes (the generated instance list). `params` may be NULL; its rounding policy is ignored
 * (nearest, like the reference), the rest applies. Asynchronous on `stream`. */
aclhip_status aclhip_decompress_all_samples(aclhip_context* context, aclhip_clip clip, const aclhip_decompress_params* params,
	void* scratch, void* out, uint64_t stride_bytes, void* stream);

/* ---- pose consumers ---------------------------------------------------------------
 * What callers of decompress_tracks do next with the local space pose, fused into the decode so that the pose buffer makes
 * one trip to HBM instead of two or three: combining an additive clip with its base (acl::apply_additive_to_base,
 * core/additive_utils.h:150-160) and local -> object space (acl::local_to_object_space,
 * compression/transform_pose_utils.h:35-50). The reference writes both in Realtime Math; DESIGN.md 4.7 says what is restated
 * and how far an x86 build of the reference can be matched (rtm::quat_normalize starts from a hardware estimate). */

/* acl::additive_clip_format8 (core/additive_utils.h:43-68) */
typedef enum aclhip_additive_format
{
	ACLHIP_ADDITIVE_NONE = 0,		/* no base: the decoded pose passes through */
	ACLHIP_ADDITIVE_RELATIVE = 1,	/* qvv_mul(additive, base) */
	ACLHIP_ADDITIVE_ADDITIVE0 = 2,	/* transform_add0: scale = additive.scale * base.scale */
	ACLHIP_ADDITIVE_ADDITIVE1 = 3	/* transform_add1: scale = (1 + additive.scale) * base.scale */
} aclhip_additive_format;

#define ACLHIP_NO_PARENT 0xFFFFFFFFu

/* Parent of every transform of a registered transform clip (track_desc_transformf::parent_index, core/track_desc.h), copied.
 * parent_indices[i] < i for every transform but the roots (sorted parent first, as local_to_object_space assumes); transform 0
 * is a root whatever parent_indices[0] says (the reference never reads it), ACLHIP_NO_PARENT marks further roots.
 * num_tracks must be the clip's. Replaces a previous hierarchy of the clip (stream ordered like aclhip_unregister_clip: the old walk
 * schedule is recycled once the launches already enqueued have completed). */
aclhip_status aclhip_set_clip_hierarchy(aclhip_context* context, aclhip_clip clip, const uint32_t* parent_indices, uint32_t num_tracks);

/* The same with the parent indices the blob itself carries (compression_metadata_settings::include_parent_track_indices /
 * include_track_descriptions; compressed_tracks::get_parent_track_index, core/impl/compressed_tracks.impl.h:175-190): the caller passes
 * nothing. ACLHIP_ERROR_NO_METADATA when the clip was registered from a blob without them. */
aclhip_status aclhip_set_clip_hierarchy_from_metadata(aclhip_context* context, aclhip_clip clip);

/* Host only (no GPU work): how aclhip_set_clip_hierarchy schedules the object space walk of a hierarchy when up to
 * `transforms_per_step` transforms can be computed at once (64 / 32 / 16 / 8 for 1 / 2 / 4 / 8 instances per workgroup): every
 * transform is scheduled after its parent, at every step the ready transforms with the longest chain of descendants go first.
 * *out_num_steps: steps the walk takes; out_steps (optional, num_tracks entries): the 1-based step of each transform, 0 for roots.
 * ACLHIP_ERROR_INVALID_ARGUMENT when a transform precedes its parent. */
aclhip_status aclhip_plan_hierarchy_walk(const uint32_t* parent_indices, uint32_t num_tracks, uint32_t transforms_per_step, uint32_t* out_steps, uint32_t* out_num_steps);

typedef struct aclhip_pose_consumers
{
	uint32_t additive_format;			/* aclhip_additive_format: how each decoded instance combines with its base pose */
	uint32_t object_space;				/* 1: convert the (combined) local pose to object space with the clip's hierarchy */
	const aclhip_clip* base_clips;		/* DEVICE [num_instances] or NULL: the base of instance i is clip base_clips[i] sampled at ... */
	const float* base_sample_times;		/* DEVICE [num_instances] ... base_sample_times[i], decoded by the same wave (same params) */
	const void* base_poses;				/* DEVICE or NULL; used when base_clips is NULL: base pose i at base_poses + i * base_pose_stride_bytes, */
	uint64_t base_pose_stride_bytes;	/* 48 bytes per transform like the output; must not alias `poses` */
	/* Blend of K clip instances (no reference function -- the reference ships the arithmetic it is made of, quat_lerp's
	 * sign bias and normalize, math/quatf.h:170-211): instance i is the weighted combination of K = num_blend_clips clip instances,
	 * clips[i] at sample_times[i] first, then blend_clips[i * (K - 1) + j] at blend_sample_times[i * (K - 1) + j], j = 0 .. K - 2,
	 * with weights blend_weights[i * K + k]. All K clips of an instance must have the same number of tracks (else refused and
	 * counted). Per transform, in this operation order, fp32, never fused (ACLHIP_BLEND_* in DESIGN.md 4.7; oracle: aclo_blend_poses):
	 *     rotation     acc = q_0 * w_0;  for k = 1 .. K-1:  dot = ((acc.x q_k.x + acc.y q_k.y) + acc.z q_k.z) + acc.w q_k.w,
	 *                  acc = (q_k * (dot < 0 ? -w_k : w_k)) + acc;  rotation = quat_normalize(acc)   (the decoder's 1 / sqrt, math/quatf.h:200-211)
	 *     translation  acc = t_0 * w_0;  acc = (t_k * w_k) + acc         scale: like the translation
	 * The weights are the caller's (normally non negative with sum 1; they are not normalized here). The blended local pose then
	 * takes the place of the decoded one: additive apply and object space follow as configured above. */
	uint32_t num_blend_clips;			/* 0 or 1: no blend; 2 .. ACLHIP_MAX_BLEND_CLIPS */
	uint32_t flags;						/* ACLHIP_CONSUMERS_* */
	const aclhip_clip* blend_clips;		/* DEVICE [num_instances * (K - 1)] */
	const float* blend_sample_times;	/* DEVICE [num_instances * (K - 1)] */
	const float* blend_weights;			/* DEVICE [num_instances * K] */
} aclhip_pose_consumers;

#define ACLHIP_MAX_BLEND_CLIPS 4u

/* aclhip_pose_consumers::flags. By default the consumers' kernels follow the reference's x86 arithmetic one IEEE operation at a time
 * (core/additive_utils.h:128-160 and compression/transform_pose_utils.h:35-50 through rtm::quat_mul / qvv_mul, math/quatf.h:135-211 for
 * the decode) and are BIT EXACT with the oracle; much of a rotation's arithmetic is then correctly rounded square roots and divisions
 * (shorter exact forms of those run for clips whose registration proved them sufficient -- the bits do not change), and
 * these kernels keep the vector ALUs busy. ACLHIP_CONSUMERS_FAST is the opt-in for callers who
 * want the poses, not the bits: object space launches (object_space != 0, no blend) then compute the same formulas with the
 * hardware's 1 ulp square root / reciprocal square root, fused multiply-adds and quat_mul_vector3 as two cross products -- in the
 * decode of the instance and of its base, the fused additive apply and the walk. Rotations stay within 2e-6 of the default's per
 * component, translations within 2e-6 of the pose's extent (tests/test_gpu_consumers.py; DESIGN.md 4.7 has the measured figures).
 * Ignored (the default arithmetic runs) for local space output and for blends; mirrored transforms keep rtm's matrix route. */
#define ACLHIP_CONSUMERS_FAST 1u

/* aclhip_decompress_tracks_batch followed by the consumers, in one kernel. `params` as for aclhip_decompress_tracks_batch but
 * restricted to what a consumer can work with -- the track_writer's own default sub-track modes, no per track rounding,
 * normalization != always -- else ACLHIP_ERROR_INVALID_ARGUMENT. Instances the kernel refuses (and counts, see
 * aclhip_get_rejected_instance_count) leave their pose untouched: unknown or scalar clips, object_space for a clip without
 * hierarchy, a base or blend clip with another number of tracks, a pose that does not fit its row. Poses are limited by the 160 KiB of
 * LDS a workgroup can use: about 3400 transforms (3100 with object space, 1700 when the base is a clip). Like every pose launch this one
 * is shaped by its batch -- pose_stride_bytes / 48 transforms, or the largest registered clip when that is smaller --
 * (ACLHIP_ERROR_INVALID_ARGUMENT when THAT does not fit): a 3 500-bone asset in the registry does not take the consumers away from the
 * 100-bone characters.
 * Asynchronous on `stream`. */
aclhip_status aclhip_decompress_poses_batch(aclhip_context* context, const aclhip_clip* clips, const float* sample_times, uint32_t num_instances,
	const aclhip_decompress_params* params, const aclhip_pose_consumers* consumers, void* poses, uint64_t pose_stride_bytes, void* stream);

/* Same with host arrays (clips, times, base clips / times / poses, output): staged through temporary device buffers, synchronous. */
aclhip_status aclhip_decompress_poses_host(aclhip_context* context, const aclhip_clip* clips, const float* sample_times, uint32_t num_instances,
	const aclhip_decompress_params* params, const aclhip_pose_consumers* consumers, void* poses, uint64_t pose_stride_bytes);

/* ---- multi-GPU ---------------------------------------------------------------------------------- */

/* Decoding never needs a collective: every GPU decodes its own contiguous shard of the instance list. Only a
 * caller that wants every rank to see every pose gathers the shards afterwards: one RCCL all-gather over xGMI.
 * `rccl_comm` is the caller's ncclComm_t (one process per GPU); `shard_poses` are this rank's `shard_bytes` bytes, `all_poses`
 * receives world_size * shard_bytes bytes in rank order (in place when shard_poses == all_poses + rank * shard_bytes).
 * DEVICE pointers; asynchronous on `stream`. librccl.so.1 is loaded on first use: ACLHIP_ERROR_DEVICE when it is absent. */
aclhip_status aclhip_all_gather_poses(aclhip_context* context, void* rccl_comm, const void* shard_poses, void* all_poses, uint64_t shard_bytes, void* stream);

/* The RCCL aclhip_all_gather_poses would call, found the same way -- a ncclAllGather already visible in the process, else a library
 * already loaded under RCCL's soname, else a fresh load of librccl.so.1 / librccl.so (the communicator was made by the RCCL of the
 * caller's process: that one must run) -- and asked for its version, without a communicator. A check to run BEFORE a multi-GPU job:
 * out_version RCCL's version code (22606 = 2.26.6), out_path the file the entry point lives in, out_how how it was found (any of the
 * three may be NULL). ACLHIP_ERROR_DEVICE when RCCL or one of the two symbols is missing. No reference counterpart. */
aclhip_status aclhip_probe_rccl(int* out_version, char* out_path, uint32_t path_capacity, char* out_how, uint32_t how_capacity);

/* Peer gather: when ONE GPU wants every pose (the one that renders), each other GPU pushes its shard straight into that GPU's
 * buffer over its own xGMI link -- the destination's seven links work concurrently and no shard travels twice -- instead of a ring
 * collective that also gives every rank every shard. One process per GPU:
 *   destination:  aclhip_peer_export_buffer(ctx, all_poses, handle)     72 opaque bytes (a HIP IPC memory handle + offset), sent to the
 *                                                                      other processes by whatever they already talk over
 *   every source: aclhip_peer_open_buffer(ctx, handle, &peer)           maps the destination's buffer (once)
 *                 aclhip_push_poses_to_peer(ctx, peer, rank * shard_bytes, shard_poses, shard_bytes, stream)   per batch, asynchronous
 *                 aclhip_peer_close_buffer(ctx, peer)
 * The destination copies its own shard with the same call on its own pointer. The caller orders the destination's reads after the
 * pushes (a barrier between the processes, or events it shares). No reference counterpart (the reference is single threaded CPU code). */
#define ACLHIP_PEER_HANDLE_BYTES 72
aclhip_status aclhip_peer_export_buffer(aclhip_context* context, void* device_buffer, uint8_t* out_handle);
aclhip_status aclhip_peer_open_buffer(aclhip_context* context, const uint8_t* handle, void** out_device_buffer);
aclhip_status aclhip_peer_close_buffer(aclhip_context* context, void* device_buffer);
aclhip_status aclhip_push_poses_to_peer(aclhip_context* context, void* peer_buffer, uint64_t offset_bytes, const void* shard_poses, uint64_t shard_bytes, void* stream);

/* Number of instances the kernels refused since the context was created (unknown clip handle, track index out of range):
 * the reference silently returns in those cases (impl/decompression.transform.h:1532-1537,1766-1768). */
aclhip_status aclhip_get_rejected_instance_count(aclhip_context* context, uint64_t* out_count);

/* Number of transforms the pose consumers combined through rtm::qvv_mul's MATRIX route (local -> object space, the `relative`
 * additive format) because a NEGATIVE scale component was involved, since the context was created: mirrored rigs, for which RTM
 * composes 3x4 matrices instead of quaternions, and so do the kernels (qvv_mul_through_matrices, bit for bit the oracle's restatement;
 * pinned by an fp64 matrix chain in tests/test_pose_consumers_oracle.py). The route is compiled into a launch only while some
 * registered clip can decode a negative scale (found at registration from defaults, constants, clip ranges and raw segments) or the
 * base is a caller's pose buffer. 0 for every rig with non-negative scales. Waits for the device like
 * aclhip_get_rejected_instance_count. */
aclhip_status aclhip_get_negative_scale_count(aclhip_context* context, uint64_t* out_count);

/* ---- measurement helpers ---------------------------------------------------------------------- */

/* Runs `repeats` launches of aclhip_decompress_tracks_batch on `stream` bracketed by HIP events recorded on that same
 * stream and returns the average milliseconds per launch (device time of the decode kernel, no host overhead). */
aclhip_status aclhip_time_decompress_tracks_batch(aclhip_context* context, const aclhip_clip* clips, const float* sample_times, uint32_t num_instances,
	const aclhip_decompress_params* params, void* poses, uint64_t pose_stride_bytes, void* stream, uint32_t repeats, float* out_ms_per_launch);

/* Same for aclhip_decompress_poses_batch. */
aclhip_status aclhip_time_decompress_poses_batch(aclhip_context* context, const aclhip_clip* clips, const float* sample_times, uint32_t num_instances,
	const aclhip_decompress_params* params, const aclhip_pose_consumers* consumers, void* poses, uint64_t pose_stride_bytes, void* stream, uint32_t repeats, float* out_ms_per_launch);

/* Name of the kernel aclhip_decompress_tracks_batch would launch for `params` with the clips registered so far, for pose rows as wide
 * as the largest registered clip (to match rocprofv3 kernel traces with bench results). */
aclhip_status aclhip_describe_tracks_kernel(aclhip_context* context, const aclhip_decompress_params* params, char* out_name, uint32_t capacity);

/* The same for the launch aclhip_decompress_tracks_batch_out makes with `output` (may be NULL) and rows of `pose_stride_bytes`: the kernel's
 * name and (optional) the wavefronts per instance -- answered by the functions the launch itself asks. */
aclhip_status aclhip_describe_tracks_launch(aclhip_context* context, const aclhip_decompress_params* params, const aclhip_output_desc* output, uint64_t pose_stride_bytes,
	char* out_name, uint32_t capacity, uint32_t* out_windows_per_instance);

/* Streams `size_bytes` of 16 byte per lane stores into `buffer` (DEVICE pointer, 16 byte aligned) `repeats` times and returns the
 * GB/s reached: the practical ceiling of a pose shaped write stream on this device, to read roofline fractions against. */
aclhip_status aclhip_measure_write_bandwidth(aclhip_context* context, void* buffer, uint64_t size_bytes, uint32_t repeats, void* stream, float* out_gb_per_second);

/* The write stream of a pose batch ALONE: one wave per pose window storing the window's rows with the pose kernels' own 1 KiB streaming
 * stores into `poses` (DEVICE pointer; num_instances rows of pose_stride_bytes, num_tracks 48 byte records each), nothing decoded,
 * at 32 / 16 / 12 / 8 resident waves per CU and with 0 / 3 / 6 dependent scalar loads pacing every wave, and the runtime's own fill of
 * the same bytes (hipMemsetAsync). Returns the best rate (and the occupancy that reached it; 0 = the runtime's fill): the BEST MEASURED
 * STORE-ONLY RATE over this buffer among those thirteen shapes -- a second denominator to read a decode's roofline fraction against
 * besides the 8 TB/s of the specification, not a bound: a decode whose stores are paced differently can come out above it.
 * OVERWRITES `poses`. Measurement aid (bench.py: roofline.best_store_only_gbps). */
aclhip_status aclhip_measure_pose_store_bandwidth(aclhip_context* context, void* poses, uint64_t pose_stride_bytes, uint32_t num_instances, uint32_t num_tracks,
	uint32_t repeats, void* stream, float* out_gb_per_second, uint32_t* out_waves_per_cu);

/* Algorithmic bytes of one batch under the compulsory-HBM model of DESIGN.md: poses written plus each distinct
 * clip's touched bytes once. `clips` is a HOST pointer here. */
aclhip_status aclhip_batch_algorithmic_bytes(const aclhip_context* context, const aclhip_clip* clips, uint32_t num_instances,
	uint64_t* out_bytes_written, uint64_t* out_distinct_clip_bytes);

#if defined(__cplusplus)
}
#endif

#endif
