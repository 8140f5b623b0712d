/* acl_oracle.h -- TEST INFRASTRUCTURE, NOT PART OF THE PRODUCT.
 *
 * Plain-C, scalar, CPU restatement of the reference's transform-track decompression path:
 *   decompression_context::seek() / decompress_tracks() / decompress_track()
 *   (/root/reference/includes/acl/decompression/decompress.h:160-172 and impl/decompression.transform.h).
 *
 * Parity status: PINNED. The restatement is checked in tests/ against
 *   (a) the reference's own unit-test vectors restated in tests/ (packing + interpolation KATs), and
 *   (b) outputs of the reference's unmodified headers compiled into oracle/_ref/libaclref.so
 *       (fixtures generated in the build container are committed under tests/golden/).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use this library.
 * Build: gcc -O2 -ffp-contract=off (the arithmetic below assumes NO fused multiply-add).
 */
#ifndef ACL_ORACLE_H
#define ACL_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { ACLO_ROUND_NONE = 0, ACLO_ROUND_FLOOR = 1, ACLO_ROUND_CEIL = 2, ACLO_ROUND_NEAREST = 3, ACLO_ROUND_PER_TRACK = 4 };
enum { ACLO_LOOP_CLAMP = 0, ACLO_LOOP_WRAP = 1, ACLO_LOOP_AS_COMPRESSED = 2 };
enum { ACLO_NORMALIZE_NEVER = 0, ACLO_NORMALIZE_LERP_ONLY = 1, ACLO_NORMALIZE_ALWAYS = 2 };
/* default_sub_track_mode (core/track_writer.h:49-74) */
enum { ACLO_DEFAULT_SKIPPED = 0, ACLO_DEFAULT_CONSTANT = 1, ACLO_DEFAULT_VARIABLE = 2, ACLO_DEFAULT_LEGACY = 3 };

/* Optional database binding: what database_context_v0 exposes to seek_v0
 * (decompression/database/impl/database_context.h:45-80). */
typedef struct aclo_database
{
	const uint8_t* clip_segment_headers;	/* runtime clip/segment header block */
	const uint8_t* bulk_data[2];			/* medium, low importance tiers; may be NULL */
} aclo_database;

typedef struct aclo_options
{
	uint8_t looping_policy;				/* ACLO_LOOP_*; AS_COMPRESSED = what the blob says */
	uint8_t normalization;				/* ACLO_NORMALIZE_*; the reference default settings use LERP_ONLY */
	uint8_t per_track_rounding;			/* decompression_settings::is_per_track_rounding_supported() */
	uint8_t default_rotation_mode;		/* ACLO_DEFAULT_* (LEGACY not allowed) */
	uint8_t default_translation_mode;	/* ACLO_DEFAULT_* (LEGACY not allowed) */
	uint8_t default_scale_mode;			/* ACLO_DEFAULT_* */
	const float* default_values;		/* CONSTANT: 12 floats (rot xyzw, trans xyz_, scale xyz_); VARIABLE: num_tracks * 12 floats; NULL = identity/0/1 */
	const uint8_t* track_rounding;		/* per track policy when seeking with PER_TRACK, num_tracks bytes */
	const aclo_database* database;		/* NULL when the clip isn't bound to a database */
} aclo_options;

/* What seek_v0 leaves in the persistent context (decompression.transform.h:533-562) */
typedef struct aclo_seek_result
{
	float sample_time;					/* clamped */
	float interpolation_alpha;
	uint32_t key_frames[2];				/* clip relative keyframes actually interpolated */
	uint32_t segment_indices[2];
	uint32_t segment_key_frames[2];		/* ordinal of the keyframe among the STORED keyframes of its data source */
	uint32_t key_frame_bit_offsets[2];
	const uint8_t* format_per_track_data[2];
	const uint8_t* segment_range_data[2];
	const uint8_t* animated_track_data[2];
	uint32_t animated_rotation_bit_size[2];
	uint32_t animated_translation_bit_size[2];
	uint8_t uses_single_segment;
} aclo_seek_result;

void aclo_default_options(aclo_options* options);

/* compressed_tracks::is_valid (core/impl/compressed_tracks.impl.h:278-301) plus the qvvf/variable-format
 * restrictions of this port. 0 = ok, otherwise an error code > 0. */
int aclo_is_valid(const void* blob, uint64_t blob_size, int check_hash);

uint32_t aclo_hash32(const void* data, uint64_t size);
uint32_t aclo_num_tracks(const void* blob);
uint32_t aclo_num_samples(const void* blob);
float aclo_sample_rate(const void* blob);
float aclo_finite_duration(const void* blob, int looping_policy);

/* find_linear_interpolation_samples_with_sample_rate (core/impl/interpolation_utils.impl.h:143-201) */
void aclo_find_linear_interpolation_samples_with_sample_rate(uint32_t num_samples, float sample_rate, float sample_time,
	int rounding_policy, int looping_policy, uint32_t* out_index0, uint32_t* out_index1, float* out_alpha);
/* find_linear_interpolation_samples_with_duration (interpolation_utils.impl.h:56-117) */
void aclo_find_linear_interpolation_samples_with_duration(uint32_t num_samples, float duration, float sample_time,
	int rounding_policy, int looping_policy, uint32_t* out_index0, uint32_t* out_index1, float* out_alpha);
/* find_linear_interpolation_alpha (interpolation_utils.impl.h:224-253) */
float aclo_find_linear_interpolation_alpha(float sample_index, uint32_t index0, uint32_t index1, int rounding_policy, int looping_policy);
float aclo_apply_rounding_policy(float alpha, int rounding_policy);

/* Bit unpackers (math/vector4_packing.h) */
void aclo_unpack_vector3_uXX(uint32_t num_bits, const uint8_t* data, uint32_t bit_offset, float out[3]);	/* :921-1035 */
void aclo_unpack_vector3_96(const uint8_t* data, uint32_t bit_offset, float out[3]);						/* :479-599 */
void aclo_unpack_vector4_128(const uint8_t* data, uint32_t bit_offset, float out[4]);						/* :59-164 */
void aclo_unpack_vector3_u48(const uint8_t* data, float out[3]);												/* :628-653 */
void aclo_unpack_vector3_u24(const uint8_t* data, float out[3]);												/* :781-818 */
/* Bit packers, only used by tests to round trip (math/vector4_packing.h:828-858, core/memory_utils.h:295-335) */
void aclo_pack_vector3_uXX(const float in[3], uint32_t num_bits, uint8_t* out_data);
void aclo_memcpy_bits(void* dest, uint64_t dest_bit_offset, const void* src, uint64_t src_bit_offset, uint64_t num_bits);

/* The reference's exhaustive uXX packing unit test (tests/sources/math/test_vector4_packing.cpp:385-465), restated. Returns the error count. */
uint32_t aclo_selftest_pack_vector3_uXX(uint32_t first_num_bits, uint32_t last_num_bits);

/* seek_v0 (decompression.transform.h:206-563). Returns 0 ok. */
int aclo_seek(const void* blob, float sample_time, int rounding_policy, const aclo_options* options, aclo_seek_result* out);

/* seek + decompress_tracks_v0 (decompression.transform.h:1526-1737).
 * out_pose: num_tracks * 12 floats (rot xyzw | trans xyz0 | scale xyz0); skipped defaults leave 'out_pose' untouched. */
int aclo_decompress_tracks(const void* blob, float sample_time, int rounding_policy, const aclo_options* options, float* out_pose);

/* seek + decompress_track_v0 (decompression.transform.h:1753-2050). out_qvv: 12 floats. */
int aclo_decompress_track(const void* blob, float sample_time, int rounding_policy, const aclo_options* options, uint32_t track_index, float* out_qvv);

/* Batch helpers: instance i = (clip blobs[clip_indices[i]], sample_times[i]); pose i goes to out + i * pose_stride_floats */
int aclo_decompress_tracks_batch(const void* const* blobs, const uint32_t* clip_indices, const float* sample_times, uint32_t count,
	int rounding_policy, const aclo_options* options, float* out, uint64_t pose_stride_floats);

/* Scalar track lists (track_type float1f / float2f / float3f / float4f / vector4f): seek_v0 + decompress_tracks_v0 /
 * decompress_track_v0 of decompression/impl/decompression.scalar.h:182-715. Components per track: 1, 2, 3, 4, 4.
 * out: num_tracks * num_components floats, tightly packed; out_value: num_components floats. options: looping_policy,
 * per_track_rounding and track_rounding are used, the rest is transform specific. Returns 1 for a transform clip. */
uint32_t aclo_scalar_num_components(const void* blob);
int aclo_scalar_decompress_tracks(const void* blob, float sample_time, int rounding_policy, const aclo_options* options, float* out);
int aclo_scalar_decompress_track(const void* blob, float sample_time, int rounding_policy, const aclo_options* options, uint32_t track_index, float* out_value);
int aclo_scalar_decompress_tracks_batch(const void* const* blobs, const uint32_t* clip_indices, const float* sample_times, uint32_t count,
	int rounding_policy, const aclo_options* options, float* out, uint64_t row_stride_floats);

/* Small utilities pinned by the reference's unit tests (tests/sources/core/test_time_utils.cpp:35-57, test_bit_manip_utils.cpp:31-60,
 * tests/sources/math/test_scalar_packing.cpp:44-140); see tests/test_oracle_kats.py */
uint32_t aclo_calculate_num_samples(float duration, float sample_rate);		/* core/impl/time_utils.impl.h:44-57 */
float aclo_calculate_duration(uint32_t num_samples, float sample_rate);			/* :59-70 */
float aclo_calculate_finite_duration(uint32_t num_samples, float sample_rate);	/* :102-112 */
uint32_t aclo_count_set_bits(uint32_t value);
uint32_t aclo_count_leading_zeros(uint32_t value);
uint32_t aclo_count_trailing_zeros(uint32_t value);
uint32_t aclo_pack_scalar_unsigned(float input, uint32_t num_bits);				/* math/scalar_packing.h:40-68 */
float aclo_unpack_scalar_unsigned(uint32_t input, uint32_t num_bits);
uint32_t aclo_pack_scalar_signed(float input, uint32_t num_bits);
float aclo_unpack_scalar_signed(uint32_t input, uint32_t num_bits);
float aclo_unpack_scalarf_32(const uint8_t* data, uint32_t bit_offset);			/* math/scalar_packing.h:71-110 */
float aclo_unpack_scalarf_uXX(uint32_t num_bits, const uint8_t* data, uint32_t bit_offset);	/* :113-160 */
uint32_t aclo_selftest_scalar_packing(uint32_t first_num_bits, uint32_t last_num_bits);

/* Pose consumers: core/additive_utils.h:128-160 and compression/transform_pose_utils.h:35-50 over poses of
 * 12 floats per transform. See acl_oracle.c for what is restated from Realtime Math and how it is pinned. */
void aclo_quat_mul(const float lhs[4], const float rhs[4], float out[4]);
void aclo_qvv_mul(const float lhs[12], const float rhs[12], float out[12]);
void aclo_apply_additive_to_base(int additive_format, const float* base_pose, const float* additive_pose, uint32_t num_transforms, float* out_pose);
void aclo_local_to_object_space(const uint32_t* parent_indices, const float* local_pose, uint32_t num_transforms, float* out_object_pose);
int aclo_decompress_poses_batch(const void* const* blobs, const uint32_t* clip_indices, const float* sample_times, uint32_t count,
	int rounding_policy, const aclo_options* options, int additive_format, const uint32_t* base_clip_indices, const float* base_sample_times,
	const uint32_t* parent_indices, uint32_t num_transforms, float* out, uint64_t pose_stride_floats);
/* Blend of K local poses (no reference function; defined in acl_oracle.c next to its code, from math/quatf.h:170-211's building blocks) */
void aclo_blend_poses(const float* const* poses, const float* weights, uint32_t num_poses, uint32_t num_transforms, float* out);
int aclo_decompress_blended_poses_batch(const void* const* blobs, const uint32_t* clip_indices, const float* sample_times, uint32_t count,
	int rounding_policy, const aclo_options* options, uint32_t num_blend_clips, const uint32_t* blend_clip_indices, const float* blend_sample_times, const float* blend_weights,
	int additive_format, const uint32_t* base_clip_indices, const float* base_sample_times,
	const uint32_t* parent_indices, uint32_t num_transforms, float* out, uint64_t pose_stride_floats);

#ifdef __cplusplus
}
#endif

#endif
