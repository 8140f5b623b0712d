/* acl_oracle.c -- TEST INFRASTRUCTURE, NOT PART OF THE PRODUCT. See acl_oracle.h.
 *
 * Scalar restatement of the reference CPU decoder. Every function cites the reference lines it follows
 * (paths relative to /root/reference/includes/acl). fp32 arithmetic is written one IEEE operation at a
 * time in the reference's order; compile with -ffp-contract=off so none of it is fused.
 */
#include "acl_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------------
 * Binary layout (core/impl/compressed_headers.h:51-131,171-197,219-325,404-439)
 * ---------------------------------------------------------------------------------------------- */
typedef struct { uint32_t size, hash; } raw_buffer_header_t;
typedef struct { uint32_t tag; uint16_t version; uint8_t algorithm_type, track_type; uint32_t num_tracks, num_samples; float sample_rate; uint32_t misc_packed; } tracks_header_t;
typedef struct { uint32_t animated_pose_bit_size, animated_rotation_bit_size, animated_translation_bit_size, segment_data; } segment_header_t;
typedef struct
{
	uint32_t num_segments, num_animated_variable_sub_tracks, num_animated_rotation_sub_tracks, num_animated_translation_sub_tracks, num_animated_scale_sub_tracks;
	uint32_t num_constant_rotation_samples, num_constant_translation_samples, num_constant_scale_samples;
	uint32_t database_header_offset, segment_headers_offset, sub_track_types_offset, constant_track_data_offset, clip_range_data_offset;
} transform_tracks_header_t;

#define TAG_COMPRESSED_TRACKS 0xac11ac11u
#define VERSION_FIRST 7
#define VERSION_V02_01_99_1 9
#define VERSION_LATEST 10
#define TRACK_TYPE_QVVF 12
#define ROTATION_FULL 0
#define ROTATION_DROP_W_FULL 2
#define ROTATION_DROP_W_VARIABLE 3
#define VECTOR_VARIABLE 1

static const tracks_header_t* get_tracks_header(const void* blob) { return (const tracks_header_t*)((const uint8_t*)blob + 8); }
static const transform_tracks_header_t* get_transform_header(const void* blob) { return (const transform_tracks_header_t*)((const uint8_t*)blob + 32); }

static int hdr_has_scale(const tracks_header_t* h) { return (h->misc_packed & 1u) != 0; }
static uint32_t hdr_default_scale(const tracks_header_t* h) { return (h->misc_packed >> 1) & 1u; }
static uint32_t hdr_scale_format(const tracks_header_t* h) { return (h->misc_packed >> 2) & 1u; }
static uint32_t hdr_translation_format(const tracks_header_t* h) { return (h->misc_packed >> 3) & 1u; }
static uint32_t hdr_rotation_format(const tracks_header_t* h) { return (h->misc_packed >> 4) & 15u; }
static int hdr_has_database(const tracks_header_t* h) { return (h->misc_packed & (1u << 8)) != 0; }
static int hdr_has_stripped_keyframes(const tracks_header_t* h) { return (h->misc_packed & (1u << 10)) != 0; }
static int hdr_is_wrap_optimized(const tracks_header_t* h) { return (h->misc_packed & (1u << 30)) != 0; }

static uint32_t load_u32(const uint8_t* p) { uint32_t v; memcpy(&v, p, 4); return v; }
static uint64_t load_u64(const uint8_t* p) { uint64_t v; memcpy(&v, p, 8); return v; }
static float load_f32(const uint8_t* p) { float v; memcpy(&v, p, 4); return v; }
static uint32_t bswap32(uint32_t v) { return (v >> 24) | ((v >> 8) & 0xFF00u) | ((v << 8) & 0xFF0000u) | (v << 24); }
static uint64_t bswap64(uint64_t v) { return ((uint64_t)bswap32((uint32_t)v) << 32) | bswap32((uint32_t)(v >> 32)); }
static float bits_to_float(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static uint32_t float_to_bits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static const uint8_t* align_ptr(const uint8_t* base, const uint8_t* p, uint32_t alignment)
{
	/* blobs are 16 byte aligned, so aligning the offset from the blob start equals aligning the pointer */
	const uint64_t offset = (uint64_t)(p - base);
	return base + ((offset + (alignment - 1)) & ~(uint64_t)(alignment - 1));
}

/* core/bit_manip_utils.h:86-202 */
static uint32_t count_set_bits(uint32_t v) { uint32_t c = 0; while (v) { v &= v - 1; c++; } return c; }
static uint32_t count_leading_zeros(uint32_t v) { uint32_t c = 0; if (v == 0) return 32; while ((v & 0x80000000u) == 0) { v <<= 1; c++; } return c; }
static uint32_t count_trailing_zeros(uint32_t v) { uint32_t c = 0; if (v == 0) return 32; while ((v & 1u) == 0) { v >>= 1; c++; } return c; }

/* core/hash.h:86-99 */
uint32_t aclo_hash32(const void* data, uint64_t size)
{
	const uint8_t* bytes = (const uint8_t*)data;
	uint32_t acc = 2166136261u;
	uint64_t i;
	for (i = 0; i < size; ++i)
		acc = (acc ^ bytes[i]) * 16777619u;
	return acc;
}

uint32_t aclo_num_tracks(const void* blob) { return get_tracks_header(blob)->num_tracks; }
uint32_t aclo_num_samples(const void* blob) { return get_tracks_header(blob)->num_samples; }
float aclo_sample_rate(const void* blob) { return get_tracks_header(blob)->sample_rate; }

/* compressed_tracks::get_looping_policy (core/impl/compressed_tracks.impl.h:140-147) */
static int resolve_looping_policy(const tracks_header_t* header, int looping_policy)
{
	if (looping_policy != ACLO_LOOP_AS_COMPRESSED)
		return looping_policy;
	if (header->version <= VERSION_FIRST)
		return ACLO_LOOP_CLAMP;
	return hdr_is_wrap_optimized(header) ? ACLO_LOOP_WRAP : ACLO_LOOP_CLAMP;
}

/* compressed_tracks::get_finite_duration (core/impl/compressed_tracks.impl.h:102-122), calculate_finite_duration (core/impl/time_utils.impl.h:102-112) */
float aclo_finite_duration(const void* blob, int looping_policy)
{
	const tracks_header_t* header = get_tracks_header(blob);
	uint32_t num_samples = header->num_samples;
	looping_policy = resolve_looping_policy(header, looping_policy);
	if (looping_policy == ACLO_LOOP_WRAP && num_samples != 0)
		num_samples++;
	if (num_samples <= 1)
		return 0.0f;
	return (float)(num_samples - 1) / header->sample_rate;
}

/* compressed_tracks::is_valid (core/impl/compressed_tracks.impl.h:278-301) */
int aclo_is_valid(const void* blob, uint64_t blob_size, int check_hash)
{
	const raw_buffer_header_t* buffer_header = (const raw_buffer_header_t*)blob;
	const tracks_header_t* header;
	if (blob == NULL)
		return 1;
	if (((uintptr_t)blob & 15u) != 0)
		return 2;	/* Invalid alignment */
	if (blob_size < 32 + sizeof(transform_tracks_header_t))
		return 3;
	header = get_tracks_header(blob);
	if (header->tag != TAG_COMPRESSED_TRACKS)
		return 4;	/* Invalid tag */
	if (header->algorithm_type != 0)
		return 5;	/* Invalid algorithm type */
	if (header->version < VERSION_FIRST || header->version > VERSION_LATEST)
		return 6;	/* Invalid algorithm version */
	if (buffer_header->size > blob_size)
		return 7;
	if (check_hash && aclo_hash32((const uint8_t*)blob + 8, buffer_header->size - 8) != buffer_header->hash)
		return 8;	/* Invalid hash */
	/* scope of this restatement: qvvf tracks in every format the reference's decoder takes (debug_transform_decompression_settings,
	 * decompression_settings.h:236-262): quatf_full / quatf_drop_w_full / quatf_drop_w_variable, vector3f_full / vector3f_variable */
	if (header->track_type != TRACK_TYPE_QVVF)
		return 9;
	if (header->num_tracks != 0 && hdr_rotation_format(header) != ROTATION_FULL && hdr_rotation_format(header) != ROTATION_DROP_W_FULL && hdr_rotation_format(header) != ROTATION_DROP_W_VARIABLE)
		return 10;
	return 0;
}

void aclo_default_options(aclo_options* options)
{
	memset(options, 0, sizeof(*options));
	options->looping_policy = ACLO_LOOP_AS_COMPRESSED;
	options->normalization = ACLO_NORMALIZE_LERP_ONLY;		/* default_transform_decompression_settings (decompression_settings.h:227) */
	options->per_track_rounding = 0;						/* decompression_settings.h:231 */
	options->default_rotation_mode = ACLO_DEFAULT_CONSTANT;	/* core/track_writer.h:161-163 */
	options->default_translation_mode = ACLO_DEFAULT_CONSTANT;
	options->default_scale_mode = ACLO_DEFAULT_LEGACY;
}

/* ------------------------------------------------------------------------------------------------
 * Interpolation helpers (core/impl/interpolation_utils.impl.h)
 * ---------------------------------------------------------------------------------------------- */
float aclo_apply_rounding_policy(float alpha, int rounding_policy)
{
	/* :261-278 */
	switch (rounding_policy)
	{
	default:
	case ACLO_ROUND_NONE:
	case ACLO_ROUND_PER_TRACK:
		return alpha;
	case ACLO_ROUND_FLOOR:
		return 0.0f;
	case ACLO_ROUND_CEIL:
		return 1.0f;
	case ACLO_ROUND_NEAREST:
		return floorf(alpha + 0.5f);
	}
}

static void find_samples_with_rate(uint32_t num_samples, float sample_rate, float sample_time, int rounding_policy, int looping_policy,
	uint32_t* out_index0, uint32_t* out_index1, float* out_alpha)
{
	/* :143-201 (shared tail of :56-117) */
	const uint32_t last_sample_index = num_samples - 1;
	float sample_index = sample_time * sample_rate;
	uint32_t sample_index0 = (uint32_t)sample_index;
	const uint32_t next_sample_index = sample_index0 + 1;
	uint32_t sample_index1;
	float interpolation_alpha;

	if (looping_policy == ACLO_LOOP_CLAMP)
		sample_index1 = next_sample_index < last_sample_index ? next_sample_index : last_sample_index;
	else
	{
		if (sample_index0 > last_sample_index)
		{
			/* sampling the repeating first sample with full weight */
			sample_index = 0.0f;
			sample_index0 = 0;
			sample_index1 = 0;
		}
		else
			sample_index1 = next_sample_index >= num_samples ? 0 : next_sample_index;
	}

	interpolation_alpha = sample_index - (float)sample_index0;

	*out_index0 = sample_index0;
	*out_index1 = sample_index1;
	*out_alpha = aclo_apply_rounding_policy(interpolation_alpha, rounding_policy);
}

void aclo_find_linear_interpolation_samples_with_sample_rate(uint32_t num_samples, float sample_rate, float sample_time,
	int rounding_policy, int looping_policy, uint32_t* out_index0, uint32_t* out_index1, float* out_alpha)
{
	find_samples_with_rate(num_samples, sample_rate, sample_time, rounding_policy, looping_policy, out_index0, out_index1, out_alpha);
}

void aclo_find_linear_interpolation_samples_with_duration(uint32_t num_samples, float duration, float sample_time,
	int rounding_policy, int looping_policy, uint32_t* out_index0, uint32_t* out_index1, float* out_alpha)
{
	/* :56-117 */
	const uint32_t last_sample_index = num_samples - 1;
	float sample_rate;
	if (duration == 0.0f)
		sample_rate = 0.0f;
	else if (looping_policy == ACLO_LOOP_CLAMP)
		sample_rate = (float)last_sample_index / duration;
	else
		sample_rate = (float)num_samples / duration;
	find_samples_with_rate(num_samples, sample_rate, sample_time, rounding_policy, looping_policy, out_index0, out_index1, out_alpha);
}

float aclo_find_linear_interpolation_alpha(float sample_index, uint32_t index0, uint32_t index1, int rounding_policy, int looping_policy)
{
	/* :224-253 */
	float interpolation_alpha;
	(void)looping_policy;

	if (rounding_policy == ACLO_ROUND_FLOOR)
		return 0.0f;
	else if (rounding_policy == ACLO_ROUND_CEIL)
		return 1.0f;
	else if (index0 == index1)
		return 0.0f;

	if (index0 < index1)
		interpolation_alpha = (sample_index - (float)index0) / (float)(index1 - index0);
	else
		interpolation_alpha = sample_index - (float)index0;

	if (rounding_policy == ACLO_ROUND_NONE || rounding_policy == ACLO_ROUND_PER_TRACK)
		return interpolation_alpha;
	return floorf(interpolation_alpha + 0.5f);
}

/* ------------------------------------------------------------------------------------------------
 * Bit unpackers (math/vector4_packing.h)
 * ---------------------------------------------------------------------------------------------- */
void aclo_unpack_vector3_uXX(uint32_t num_bits, const uint8_t* data, uint32_t bit_offset, float out[3])
{
	/* :921-1035: three unaligned big-endian 32 bit windows, shift, mask, int -> float, scale by 1/(2^n - 1) */
	const uint32_t bit_shift = 32 - num_bits;
	const uint32_t mask = (1u << num_bits) - 1;
	const float inv_max_value = num_bits == 0 ? 1.0f : (1.0f / (float)((1 << num_bits) - 1));
	int c;
	for (c = 0; c < 3; ++c)
	{
		const uint32_t byte_offset = bit_offset / 8;
		const uint32_t window = bswap32(load_u32(data + byte_offset));
		const uint32_t value = (window >> (bit_shift - (bit_offset % 8))) & mask;
		out[c] = (float)value * inv_max_value;
		bit_offset += num_bits;
	}
}

void aclo_unpack_vector3_96(const uint8_t* data, uint32_t bit_offset, float out[3])
{
	/* :479-599: three big-endian IEEE floats starting at an arbitrary bit */
	const uint32_t byte_offset = bit_offset / 8;
	const uint32_t shift_offset = bit_offset % 8;
	int c;
	for (c = 0; c < 3; ++c)
	{
		uint64_t v = bswap64(load_u64(data + byte_offset + 4 * (uint32_t)c));
		v <<= shift_offset;
		v >>= 32;
		out[c] = bits_to_float((uint32_t)v);
	}
}

void aclo_unpack_vector4_128(const uint8_t* data, uint32_t bit_offset, float out[4])
{
	/* unpack_vector4_128_unsafe, math/vector4_packing.h:59-164: four big-endian IEEE floats starting at an arbitrary bit */
	const uint32_t byte_offset = bit_offset / 8;
	const uint32_t shift_offset = bit_offset % 8;
	int c;
	for (c = 0; c < 4; ++c)
	{
		uint64_t v = bswap64(load_u64(data + byte_offset + 4 * (uint32_t)c));
		v <<= shift_offset;
		v >>= 32;
		out[c] = bits_to_float((uint32_t)v);
	}
}

void aclo_unpack_vector3_u48(const uint8_t* data, float out[3])
{
	/* :628-653: three little-endian u16, scaled by 1/65535 */
	int c;
	for (c = 0; c < 3; ++c)
	{
		const uint32_t v = (uint32_t)data[c * 2] | ((uint32_t)data[c * 2 + 1] << 8);
		out[c] = (float)v * (1.0f / 65535.0f);
	}
}

void aclo_unpack_vector3_u24(const uint8_t* data, float out[3])
{
	/* :781-818: three u8, scaled by 1/255 */
	int c;
	for (c = 0; c < 3; ++c)
		out[c] = (float)data[c] * (1.0f / 255.0f);
}

void aclo_memcpy_bits(void* dest, uint64_t dest_bit_offset, const void* src, uint64_t src_bit_offset, uint64_t num_bits)
{
	/* core/memory_utils.h:295-335: MSB-first bit copy */
	uint8_t* d = (uint8_t*)dest;
	const uint8_t* s = (const uint8_t*)src;
	uint64_t i;
	for (i = 0; i < num_bits; ++i)
	{
		const uint64_t sb = src_bit_offset + i, db = dest_bit_offset + i;
		const uint8_t bit = (uint8_t)((s[sb >> 3] >> (7 - (sb & 7))) & 1u);
		d[db >> 3] = (uint8_t)((d[db >> 3] & ~(0x80u >> (db & 7))) | (bit << (7 - (db & 7))));
	}
}

void aclo_pack_vector3_uXX(const float in[3], uint32_t num_bits, uint8_t* out_data)
{
	/* math/vector4_packing.h:828-858 with pack_scalar_unsigned (math/scalar_packing.h:42-48): round half away from zero */
	const float max_value = (float)((1u << num_bits) - 1);
	uint64_t bit_offset = 0;
	int c;
	memset(out_data, 0, 16);
	for (c = 0; c < 3; ++c)
	{
		const uint32_t q = (uint32_t)floorf(in[c] * max_value + 0.5f);
		const uint32_t be = bswap32(q << (32 - num_bits));
		aclo_memcpy_bits(out_data, bit_offset, &be, 0, num_bits);
		bit_offset += num_bits;
	}
}

/* ------------------------------------------------------------------------------------------------
 * seek_v0 (decompression/impl/decompression.transform.h:206-563)
 * ---------------------------------------------------------------------------------------------- */
static void get_segment_data(const void* blob, const transform_tracks_header_t* th, const segment_header_t* sh,
	const uint8_t** out_format, const uint8_t** out_range, const uint8_t** out_animated)
{
	/* core/impl/compressed_headers.h:309-324 */
	const uint8_t* base = (const uint8_t*)blob;
	const uint8_t* format_per_track_data = (const uint8_t*)th + sh->segment_data;
	const uint8_t* range_data = align_ptr(base, format_per_track_data + th->num_animated_variable_sub_tracks, 2);
	const uint32_t range_data_size = th->num_segments > 1 ? 6 * th->num_animated_variable_sub_tracks : 0;
	*out_format = format_per_track_data;
	*out_range = range_data;
	*out_animated = align_ptr(base, range_data + range_data_size, 4);
}

int aclo_seek(const void* blob, float sample_time, int rounding_policy, const aclo_options* options, aclo_seek_result* out)
{
	const tracks_header_t* header = get_tracks_header(blob);
	const transform_tracks_header_t* th = get_transform_header(blob);
	const uint8_t* tbase = (const uint8_t*)th;
	const int looping_policy = resolve_looping_policy(header, options->looping_policy);
	const float clip_duration = aclo_finite_duration(blob, looping_policy);
	const uint32_t num_segments = th->num_segments;
	const int has_database = hdr_has_database(header);
	const aclo_database* db = options->database;
	const int has_stripped_keyframes = has_database || hdr_has_stripped_keyframes(header);
	const uint32_t segment_header_size = has_stripped_keyframes ? 20u : 16u;
	const uint8_t* segment_headers = tbase + th->segment_headers_offset;
	uint32_t key_frame0, key_frame1, segment_key_frame0, segment_key_frame1;
	uint32_t segment_index0 = 0, segment_index1 = 0;
	float alpha;
	const uint8_t* db_animated_track_data0 = NULL;
	const uint8_t* db_animated_track_data1 = NULL;
	const segment_header_t* segment_header0;
	const segment_header_t* segment_header1;

	memset(out, 0, sizeof(*out));
	if (header->num_tracks == 0)
		return 1;

	/* :215-216 clamp (scalar_clamp = min(max(t, 0), duration)) */
	sample_time = sample_time > 0.0f ? sample_time : 0.0f;
	sample_time = sample_time < clip_duration ? sample_time : clip_duration;

	/* :240 */
	find_samples_with_rate(header->num_samples, header->sample_rate, sample_time, rounding_policy, looping_policy, &key_frame0, &key_frame1, &alpha);

	if (num_segments == 1)
	{
		/* :267-371 */
		if (has_stripped_keyframes)
		{
			uint32_t sample_indices0 = load_u32(segment_headers + 16);
			uint32_t sample_indices1;
			const float sample_index = alpha + (float)key_frame0;
			uint64_t medium0 = 0, low0 = 0;
			uint32_t candidates;

			if (db != NULL)
			{
				const uint8_t* tracks_db_header = tbase + th->database_header_offset;
				const uint8_t* db_clip_header = db->clip_segment_headers + load_u32(tracks_db_header);
				const uint8_t* db_segment_headers = db_clip_header + 8;
				medium0 = load_u64(db_segment_headers + 0);
				low0 = load_u64(db_segment_headers + 8);
				sample_indices0 |= (uint32_t)medium0;
				sample_indices0 |= (uint32_t)low0;
			}

			candidates = sample_indices0 & (0xFFFFFFFFu << (31 - key_frame0));
			key_frame0 = 31 - count_trailing_zeros(candidates);
			candidates = sample_indices0 & (0xFFFFFFFFu >> key_frame1);
			key_frame1 = count_leading_zeros(candidates);

			alpha = aclo_find_linear_interpolation_alpha(sample_index, key_frame0, key_frame1, ACLO_ROUND_NONE, looping_policy);

			sample_indices0 = load_u32(segment_headers + 16);
			sample_indices1 = sample_indices0;

			if (db != NULL)
			{
				const uint64_t bit0 = (uint64_t)1 << (31 - key_frame0);
				const uint64_t bit1 = (uint64_t)1 << (31 - key_frame1);
				if ((medium0 & bit0) != 0) { sample_indices0 = (uint32_t)medium0; db_animated_track_data0 = db->bulk_data[0] + (uint32_t)(medium0 >> 32); }
				else if ((low0 & bit0) != 0) { sample_indices0 = (uint32_t)low0; db_animated_track_data0 = db->bulk_data[1] + (uint32_t)(low0 >> 32); }
				if ((medium0 & bit1) != 0) { sample_indices1 = (uint32_t)medium0; db_animated_track_data1 = db->bulk_data[0] + (uint32_t)(medium0 >> 32); }
				else if ((low0 & bit1) != 0) { sample_indices1 = (uint32_t)low0; db_animated_track_data1 = db->bulk_data[1] + (uint32_t)(low0 >> 32); }
			}

			segment_key_frame0 = count_set_bits(~(0xFFFFFFFFu >> key_frame0) & sample_indices0);
			segment_key_frame1 = count_set_bits(~(0xFFFFFFFFu >> key_frame1) & sample_indices1);
		}
		else
		{
			segment_key_frame0 = key_frame0;
			segment_key_frame1 = key_frame1;
		}
	}
	else
	{
		/* :372-521 */
		const uint8_t* segment_start_indices = tbase + 52;
		const uint32_t approx_num_samples_per_segment = header->num_samples / num_segments;
		const uint32_t approx_segment_index = key_frame0 / approx_num_samples_per_segment;
		const uint32_t start_segment_index = approx_segment_index > 0 ? (approx_segment_index - 1) : 0;
		const uint32_t end_segment_index = start_segment_index + 4;
		uint32_t segment_index;

		for (segment_index = start_segment_index; segment_index < end_segment_index; ++segment_index)
		{
			if (key_frame0 < load_u32(segment_start_indices + 4 * segment_index))
			{
				segment_index0 = segment_index - 1;
				if (key_frame1 == 0)	/* wrapping is supported and we wrapped: use the first segment */
					segment_index1 = 0;
				else
					segment_index1 = key_frame1 < load_u32(segment_start_indices + 4 * segment_index) ? segment_index0 : segment_index;
				break;
			}
		}

		segment_key_frame0 = key_frame0 - load_u32(segment_start_indices + 4 * segment_index0);
		segment_key_frame1 = key_frame1 - load_u32(segment_start_indices + 4 * segment_index1);

		if (has_stripped_keyframes)
		{
			const uint8_t* h0 = segment_headers + 20 * segment_index0;
			const uint8_t* h1 = segment_headers + 20 * segment_index1;
			uint32_t sample_indices0 = load_u32(h0 + 16);
			uint32_t sample_indices1 = load_u32(h1 + 16);
			const float sample_index = alpha + (float)key_frame0;
			uint64_t medium0 = 0, medium1 = 0, low0 = 0, low1 = 0;
			uint32_t candidates, clip_key_frame0, clip_key_frame1;

			if (db != NULL)
			{
				const uint8_t* tracks_db_header = tbase + th->database_header_offset;
				const uint8_t* db_clip_header = db->clip_segment_headers + load_u32(tracks_db_header);
				const uint8_t* db_segment_headers = db_clip_header + 8;
				medium0 = load_u64(db_segment_headers + 16 * segment_index0 + 0);
				low0 = load_u64(db_segment_headers + 16 * segment_index0 + 8);
				sample_indices0 |= (uint32_t)medium0;
				sample_indices0 |= (uint32_t)low0;
				medium1 = load_u64(db_segment_headers + 16 * segment_index1 + 0);
				low1 = load_u64(db_segment_headers + 16 * segment_index1 + 8);
				sample_indices1 |= (uint32_t)medium1;
				sample_indices1 |= (uint32_t)low1;
			}

			candidates = sample_indices0 & (0xFFFFFFFFu << (31 - segment_key_frame0));
			segment_key_frame0 = 31 - count_trailing_zeros(candidates);
			candidates = sample_indices1 & (0xFFFFFFFFu >> segment_key_frame1);
			segment_key_frame1 = count_leading_zeros(candidates);

			clip_key_frame0 = load_u32(segment_start_indices + 4 * segment_index0) + segment_key_frame0;
			clip_key_frame1 = load_u32(segment_start_indices + 4 * segment_index1) + segment_key_frame1;
			key_frame0 = clip_key_frame0;
			key_frame1 = clip_key_frame1;

			alpha = aclo_find_linear_interpolation_alpha(sample_index, clip_key_frame0, clip_key_frame1, ACLO_ROUND_NONE, looping_policy);

			sample_indices0 = load_u32(h0 + 16);
			sample_indices1 = load_u32(h1 + 16);

			if (db != NULL)
			{
				const uint64_t bit0 = (uint64_t)1 << (31 - segment_key_frame0);
				const uint64_t bit1 = (uint64_t)1 << (31 - segment_key_frame1);
				if ((medium0 & bit0) != 0) { sample_indices0 = (uint32_t)medium0; db_animated_track_data0 = db->bulk_data[0] + (uint32_t)(medium0 >> 32); }
				else if ((low0 & bit0) != 0) { sample_indices0 = (uint32_t)low0; db_animated_track_data0 = db->bulk_data[1] + (uint32_t)(low0 >> 32); }
				if ((medium1 & bit1) != 0) { sample_indices1 = (uint32_t)medium1; db_animated_track_data1 = db->bulk_data[0] + (uint32_t)(medium1 >> 32); }
				else if ((low1 & bit1) != 0) { sample_indices1 = (uint32_t)low1; db_animated_track_data1 = db->bulk_data[1] + (uint32_t)(low1 >> 32); }
			}

			segment_key_frame0 = count_set_bits(~(0xFFFFFFFFu >> segment_key_frame0) & sample_indices0);
			segment_key_frame1 = count_set_bits(~(0xFFFFFFFFu >> segment_key_frame1) & sample_indices1);
		}
	}

	segment_header0 = (const segment_header_t*)(segment_headers + (uint64_t)segment_header_size * segment_index0);
	segment_header1 = (const segment_header_t*)(segment_headers + (uint64_t)segment_header_size * segment_index1);

	/* :530-562 */
	out->sample_time = sample_time;
	out->interpolation_alpha = alpha;
	out->key_frames[0] = key_frame0;
	out->key_frames[1] = key_frame1;
	out->segment_indices[0] = segment_index0;
	out->segment_indices[1] = segment_index1;
	out->segment_key_frames[0] = segment_key_frame0;
	out->segment_key_frames[1] = segment_key_frame1;
	out->uses_single_segment = segment_header0 == segment_header1;

	get_segment_data(blob, th, segment_header0, &out->format_per_track_data[0], &out->segment_range_data[0], &out->animated_track_data[0]);
	get_segment_data(blob, th, segment_header1, &out->format_per_track_data[1], &out->segment_range_data[1], &out->animated_track_data[1]);

	if (has_database)
	{
		if (db_animated_track_data0 != NULL)
			out->animated_track_data[0] = db_animated_track_data0;
		if (db_animated_track_data1 != NULL)
			out->animated_track_data[1] = db_animated_track_data1;
	}

	out->key_frame_bit_offsets[0] = segment_key_frame0 * segment_header0->animated_pose_bit_size;
	out->key_frame_bit_offsets[1] = segment_key_frame1 * segment_header1->animated_pose_bit_size;
	out->animated_rotation_bit_size[0] = segment_header0->animated_rotation_bit_size;
	out->animated_rotation_bit_size[1] = segment_header1->animated_rotation_bit_size;
	out->animated_translation_bit_size[0] = segment_header0->animated_translation_bit_size;
	out->animated_translation_bit_size[1] = segment_header1->animated_translation_bit_size;
	return 0;
}

/* ------------------------------------------------------------------------------------------------
 * Arithmetic core (math/quatf.h, decompression/impl/animated_track_cache.transform.h)
 * ---------------------------------------------------------------------------------------------- */
/* math/quatf.h:135-147: w = sqrt(|((1 - x*x) - y*y) - z*z|) */
static float quat_from_positive_w(float x, float y, float z)
{
	float result = 1.0f - (x * x);
	result = result - (y * y);
	result = result - (z * z);
	return sqrtf(fabsf(result));
}

/* math/quatf.h:200-211 */
static void quat_normalize(float q[4])
{
	float dot = q[0] * q[0];
	float len, inv_len;
	dot = (q[1] * q[1]) + dot;
	dot = (q[2] * q[2]) + dot;
	dot = (q[3] * q[3]) + dot;
	len = sqrtf(dot);
	inv_len = 1.0f / len;
	q[0] = q[0] * inv_len;
	q[1] = q[1] * inv_len;
	q[2] = q[2] * inv_len;
	q[3] = q[3] * inv_len;
}

/* math/quatf.h:170-196 */
static void quat_lerp_no_normalization(const float q0[4], const float q1[4], float alpha, float out[4])
{
	float dot = q0[0] * q1[0];
	uint32_t bias;
	int c;
	dot = (q0[1] * q1[1]) + dot;
	dot = (q0[2] * q1[2]) + dot;
	dot = (q0[3] * q1[3]) + dot;
	bias = float_to_bits(dot) & 0x80000000u;
	for (c = 0; c < 4; ++c)
	{
		const float end_with_bias = bits_to_float(float_to_bits(q1[c]) ^ bias);
		const float start_part = q0[c] - (q0[c] * alpha);	/* vector_neg_mul_sub(start, alpha, start) */
		out[c] = (end_with_bias * alpha) + start_part;		/* vector_mul_add(end, alpha, start_part) */
	}
}

/* rtm::vector_lerp, stable form */
static void vector_lerp3(const float v0[3], const float v1[3], float alpha, float out[3])
{
	int c;
	for (c = 0; c < 3; ++c)
	{
		const float start_part = v0[c] - (v0[c] * alpha);
		out[c] = (v1[c] * alpha) + start_part;
	}
}

typedef struct
{
	const tracks_header_t* header;
	const transform_tracks_header_t* th;
	const aclo_seek_result* seek;
	uint32_t raw_num_bits;		/* 31 from v02_01_99_1 on, 32 before (animated_track_cache.transform.h:523) */
	int has_segments;
	/* the clip's packed formats (decompression.transform.h:95-116). Only the VARIABLE formats carry per track metadata (one format byte per
	 * segment), segment ranges and clip ranges; the full formats store every sample as 96 (quatf_drop_w_full, vector3f_full) or 128
	 * (quatf_full) raw bits and nothing else (animated_track_cache.transform.h:1254-1300) */
	int rotations_variable, rotations_full, translations_variable, scales_variable;
} decode_ctx_t;

/* Bits a sub-track occupies per component in the animated pose (count_animated_group_bit_size, animated_track_cache.transform.h:1105-1192) */
static uint32_t stored_bits(const decode_ctx_t* ctx, uint32_t num_bits) { return num_bits == ctx->raw_num_bits ? 32u : num_bits; }

/* One animated rotation sample of one key frame, whole-pose flavour: unpack_animated_quat (:515-687) + remap_segment_range_data4 (:302-350)
 * + remap_clip_range_data4 (:391-466). 'index' is the ordinal among animated rotations, 'bit_offset' the sub-track's bit position. */
static void unpack_rotation_sample(const decode_ctx_t* ctx, int key, uint32_t index, uint32_t bit_offset, float out_xyz[4])
{
	const uint32_t group = index / 4, lane = index % 4;
	const uint32_t num_rotations = ctx->th->num_animated_rotation_sub_tracks;
	const uint32_t group_size = (num_rotations - group * 4) < 4 ? (num_rotations - group * 4) : 4;
	const uint8_t* format_per_track_data = ctx->seek->format_per_track_data[key];
	const uint8_t* segment_range_data = ctx->seek->segment_range_data[key] + (uint64_t)group * 24 + lane;
	const uint8_t* clip_range_data = (const uint8_t*)ctx->th + ctx->th->clip_range_data_offset + (uint64_t)group * 96 + (uint64_t)lane * 4;
	uint32_t num_bits;
	int ignore_segment, ignore_clip, c;

	if (!ctx->rotations_variable)
	{
		/* :608-620 / :796-806: raw samples, no metadata, no range reduction (the masks are zero and the remaps are not called, :1384-1412) */
		if (ctx->rotations_full)
			aclo_unpack_vector4_128(ctx->seek->animated_track_data[key], bit_offset, out_xyz);
		else
			aclo_unpack_vector3_96(ctx->seek->animated_track_data[key], bit_offset, out_xyz);
		return;
	}

	num_bits = format_per_track_data[index];
	if (num_bits == 0)
	{
		/* constant in this segment: 16 bit sample hidden in the segment range bytes, hi/lo split across the SOA rows (:552-588) */
		const uint32_t x = ((uint32_t)segment_range_data[0] << 8) | segment_range_data[4];
		const uint32_t y = ((uint32_t)segment_range_data[8] << 8) | segment_range_data[12];
		const uint32_t z = ((uint32_t)segment_range_data[16] << 8) | segment_range_data[20];
		out_xyz[0] = (float)x * (1.0f / 65535.0f);
		out_xyz[1] = (float)y * (1.0f / 65535.0f);
		out_xyz[2] = (float)z * (1.0f / 65535.0f);
		ignore_segment = 1;
		ignore_clip = 0;
	}
	else if (num_bits == ctx->raw_num_bits)
	{
		aclo_unpack_vector3_96(ctx->seek->animated_track_data[key], bit_offset, out_xyz);
		ignore_segment = 1;
		ignore_clip = 1;
	}
	else
	{
		aclo_unpack_vector3_uXX(num_bits, ctx->seek->animated_track_data[key], bit_offset, out_xyz);
		ignore_segment = 0;
		ignore_clip = 0;
	}

	if (ctx->has_segments)
	{
		/* ignored lanes still go through the multiply-add with extent 1 and min 0 (:316-349) */
		for (c = 0; c < 3; ++c)
		{
			const float range_min = ignore_segment ? 0.0f : (float)segment_range_data[c * 4] * (1.0f / 255.0f);
			const float range_extent = ignore_segment ? 1.0f : (float)segment_range_data[12 + c * 4] * (1.0f / 255.0f);
			out_xyz[c] = (out_xyz[c] * range_extent) + range_min;
		}
	}

	for (c = 0; c < 3; ++c)
	{
		const float range_min = ignore_clip ? 0.0f : load_f32(clip_range_data + (uint64_t)group_size * 4 * (uint32_t)c);
		const float range_extent = ignore_clip ? 1.0f : load_f32(clip_range_data + (uint64_t)group_size * 4 * (3 + (uint32_t)c));
		out_xyz[c] = (out_xyz[c] * range_extent) + range_min;
	}
}

/* One animated translation/scale sample of one key frame: unpack_animated_vector3 (:871-990).
 * 'format_index' indexes format_per_track_data, 'range_index' is the ordinal within the translation (or scale) sub-tracks,
 * range bases point at the first translation (or scale) entry. */
static void unpack_vector3_sample(const decode_ctx_t* ctx, int key, uint32_t format_index, const uint8_t* segment_range_base, const uint8_t* clip_range_base,
	uint32_t range_index, uint32_t bit_offset, float out_xyz[3])
{
	uint32_t num_bits;
	const uint8_t* segment_range_data = segment_range_base + (uint64_t)range_index * 6;
	const uint8_t* clip_range_data = clip_range_base + (uint64_t)range_index * 24;
	int ignore_segment, ignore_clip, c;

	if (format_index == 0xFFFFFFFFu)
	{
		/* vector3f_full (:921-926, :1071): three raw floats, no ranges */
		aclo_unpack_vector3_96(ctx->seek->animated_track_data[key], bit_offset, out_xyz);
		return;
	}

	num_bits = ctx->seek->format_per_track_data[key][format_index];
	if (num_bits == 0)
	{
		aclo_unpack_vector3_u48(segment_range_data, out_xyz);
		ignore_segment = 1;
		ignore_clip = 0;
	}
	else if (num_bits == ctx->raw_num_bits)
	{
		aclo_unpack_vector3_96(ctx->seek->animated_track_data[key], bit_offset, out_xyz);
		ignore_segment = 1;
		ignore_clip = 1;
	}
	else
	{
		aclo_unpack_vector3_uXX(num_bits, ctx->seek->animated_track_data[key], bit_offset, out_xyz);
		ignore_segment = 0;
		ignore_clip = 0;
	}

	if (ctx->has_segments && !ignore_segment)
	{
		float range_min[3], range_extent[3];
		aclo_unpack_vector3_u24(segment_range_data, range_min);
		aclo_unpack_vector3_u24(segment_range_data + 3, range_extent);
		for (c = 0; c < 3; ++c)
			out_xyz[c] = (out_xyz[c] * range_extent[c]) + range_min[c];
	}

	if (!ignore_clip)
	{
		for (c = 0; c < 3; ++c)
			out_xyz[c] = (out_xyz[c] * load_f32(clip_range_data + 12 + 4 * c)) + load_f32(clip_range_data + 4 * c);
	}
}

/* sub-track class of track 'index' (core/impl/compressed_headers.h:214-224) */
static uint32_t sub_track_type(const uint8_t* types, uint32_t index)
{
	return (load_u32(types + 4 * (index / 16)) >> ((15 - (index % 16)) * 2)) & 3u;
}

static void write_default(const aclo_options* options, int kind, uint8_t mode, uint32_t track_index, const tracks_header_t* header, float* qvv)
{
	/* decompression.transform.h:575-675, 883-985, 1203-1310, 1643-1680 */
	float* dst = qvv + kind * 4;
	if (mode == ACLO_DEFAULT_SKIPPED)
		return;
	if (mode == ACLO_DEFAULT_VARIABLE && options->default_values != NULL)
	{
		const float* src = options->default_values + (uint64_t)track_index * 12 + kind * 4;
		dst[0] = src[0]; dst[1] = src[1]; dst[2] = src[2]; dst[3] = kind == 0 ? src[3] : 0.0f;
		return;
	}
	if (mode == ACLO_DEFAULT_CONSTANT && options->default_values != NULL)
	{
		const float* src = options->default_values + kind * 4;
		dst[0] = src[0]; dst[1] = src[1]; dst[2] = src[2]; dst[3] = kind == 0 ? src[3] : 0.0f;
		return;
	}
	if (kind == 0) { dst[0] = 0.0f; dst[1] = 0.0f; dst[2] = 0.0f; dst[3] = 1.0f; }
	else if (kind == 1) { dst[0] = 0.0f; dst[1] = 0.0f; dst[2] = 0.0f; dst[3] = 0.0f; }
	else
	{
		/* legacy scale = the blob's default scale bit, otherwise 1 (core/track_writer.h:169) */
		const float scale = mode == ACLO_DEFAULT_LEGACY ? (float)hdr_default_scale(header) : 1.0f;
		dst[0] = scale; dst[1] = scale; dst[2] = scale; dst[3] = 0.0f;
	}
}

/* Shared worker: decodes every track (track_filter < 0) or only one. 'single' selects decompress_track_v0's variations. */
static int decode_pose(const void* blob, float sample_time, int rounding_policy, const aclo_options* options, int track_filter, float* out)
{
	const tracks_header_t* header = get_tracks_header(blob);
	const transform_tracks_header_t* th = get_transform_header(blob);
	const uint8_t* tbase = (const uint8_t*)th;
	const uint32_t num_tracks = header->num_tracks;
	const uint32_t num_entries = (num_tracks + 15) / 16;
	const int has_scale = hdr_has_scale(header);
	const uint8_t* rotation_types = tbase + th->sub_track_types_offset;
	const uint8_t* translation_types = rotation_types + 4 * num_entries;
	const uint8_t* scale_types = translation_types + 4 * num_entries;
	/* the formats decide what exists per sub-track (constant_track_cache.transform.h:102-110, animated_track_cache.transform.h:1254-1300) */
	const int rotations_variable = hdr_rotation_format(header) == ROTATION_DROP_W_VARIABLE;
	const int rotations_full = hdr_rotation_format(header) == ROTATION_FULL;
	const int translations_variable = hdr_translation_format(header) == VECTOR_VARIABLE;
	const int scales_variable = hdr_scale_format(header) == VECTOR_VARIABLE;
	const uint8_t* constant_rotations = tbase + th->constant_track_data_offset;
	const uint8_t* constant_translations = constant_rotations + (rotations_full ? 16u : 12u) * (uint64_t)th->num_constant_rotation_samples;
	const uint8_t* constant_scales = constant_translations + 12 * (uint64_t)th->num_constant_translation_samples;
	/* entries in front of the translations / scales in the per track format bytes and in the segment range data: only variable sub-tracks have any */
	const uint32_t num_rotations_padded = rotations_variable ? ((th->num_animated_rotation_sub_tracks + 3) & ~3u) : 0u;
	const uint32_t num_translation_entries = translations_variable ? th->num_animated_translation_sub_tracks : 0u;
	const uint8_t* clip_range_rotations = tbase + th->clip_range_data_offset;
	const uint8_t* clip_range_translations = clip_range_rotations + (rotations_variable ? 24u : 0u) * (uint64_t)th->num_animated_rotation_sub_tracks;
	const uint8_t* clip_range_scales = clip_range_translations + (translations_variable ? 24u : 0u) * (uint64_t)th->num_animated_translation_sub_tracks;
	const int single = track_filter >= 0;
	aclo_seek_result seek;
	decode_ctx_t ctx;
	uint32_t track_index;
	uint32_t constant_counts[3] = { 0, 0, 0 };
	uint32_t animated_counts[3] = { 0, 0, 0 };
	uint32_t bit_offsets[3][2];
	int key, kind;

	if (num_tracks == 0)
		return 1;
	if (single && (uint32_t)track_filter >= num_tracks)
		return 2;
	if (aclo_seek(blob, sample_time, rounding_policy, options, &seek) != 0)
		return 1;

	ctx.header = header;
	ctx.th = th;
	ctx.seek = &seek;
	ctx.raw_num_bits = header->version >= VERSION_V02_01_99_1 ? 31u : 32u;
	ctx.has_segments = th->num_segments > 1;
	ctx.rotations_variable = rotations_variable;
	ctx.rotations_full = rotations_full;
	ctx.translations_variable = translations_variable;
	ctx.scales_variable = scales_variable;

	/* animated_track_cache_v0::initialize (animated_track_cache.transform.h:1223-1314) */
	for (key = 0; key < 2; ++key)
	{
		bit_offsets[0][key] = seek.key_frame_bit_offsets[key];
		bit_offsets[1][key] = bit_offsets[0][key] + seek.animated_rotation_bit_size[key];
		bit_offsets[2][key] = bit_offsets[1][key] + seek.animated_translation_bit_size[key];
	}

	for (track_index = 0; track_index < num_tracks; ++track_index)
	{
		const int wanted = !single || track_index == (uint32_t)track_filter;
		float* qvv = single ? out : out + (uint64_t)track_index * 12;

		for (kind = 0; kind < 3; ++kind)
		{
			const uint8_t* types = kind == 0 ? rotation_types : (kind == 1 ? translation_types : scale_types);
			const uint32_t type = (kind == 2 && !has_scale) ? 0u : sub_track_type(types, track_index);
			const uint8_t default_mode = kind == 0 ? options->default_rotation_mode : (kind == 1 ? options->default_translation_mode : options->default_scale_mode);

			if (type == 0)
			{
				if (wanted)
					write_default(options, kind, default_mode, track_index, header, qvv);
			}
			else if (type == 1)
			{
				const uint32_t index = constant_counts[kind]++;
				if (!wanted)
					continue;

				if (kind == 0)
				{
					/* constant_track_cache_v0::unpack_rotation_group (constant_track_cache.transform.h:113-205): SOA groups of 4, last group unpadded */
					const uint32_t group = index / 4, lane = index % 4;
					const uint32_t left = th->num_constant_rotation_samples - group * 4;
					const uint32_t group_size = left < 4 ? left : 4;
					const uint8_t* group_data = constant_rotations + (uint64_t)group * 48;
					float q[4];
					if (rotations_full)
					{
						/* unpack_quat_128 (:136-149, :220-224): AOS xyzw, stored as is -- never normalized, whatever the policy */
						const uint8_t* src = constant_rotations + (uint64_t)index * 16;
						q[0] = load_f32(src + 0); q[1] = load_f32(src + 4); q[2] = load_f32(src + 8); q[3] = load_f32(src + 12);
					}
					else
					{
						q[0] = load_f32(group_data + 4 * (uint64_t)(group_size * 0 + lane));
						q[1] = load_f32(group_data + 4 * (uint64_t)(group_size * 1 + lane));
						q[2] = load_f32(group_data + 4 * (uint64_t)(group_size * 2 + lane));
						q[3] = quat_from_positive_w(q[0], q[1], q[2]);
						if (options->normalization == ACLO_NORMALIZE_ALWAYS)
							quat_normalize(q);
					}
					qvv[0] = q[0]; qvv[1] = q[1]; qvv[2] = q[2]; qvv[3] = q[3];
				}
				else
				{
					/* consume_translation / consume_scale (constant_track_cache.transform.h:297-302,328-333) */
					const uint8_t* src = (kind == 1 ? constant_translations : constant_scales) + (uint64_t)index * 12;
					qvv[kind * 4 + 0] = load_f32(src + 0);
					qvv[kind * 4 + 1] = load_f32(src + 4);
					qvv[kind * 4 + 2] = load_f32(src + 8);
					qvv[kind * 4 + 3] = 0.0f;
				}
			}
			else
			{
				const uint32_t index = animated_counts[kind]++;
				const int kind_variable = kind == 0 ? rotations_variable : (kind == 1 ? translations_variable : scales_variable);
				const uint32_t format_index = !kind_variable ? 0xFFFFFFFFu : (kind == 0 ? index : (kind == 1 ? num_rotations_padded + index : num_rotations_padded + num_translation_entries + index));
				uint32_t sample_bit_offsets[2];
				float alpha = seek.interpolation_alpha;
				int policy = ACLO_ROUND_NONE;

				for (key = 0; key < 2; ++key)
				{
					sample_bit_offsets[key] = bit_offsets[kind][key];
					/* full formats: 96 bits per sample, 128 for quatf_full (:608-620, skip_rotation_groups :1697-1706) */
					bit_offsets[kind][key] += kind_variable ? stored_bits(&ctx, seek.format_per_track_data[key][format_index]) * 3 : ((kind == 0 && rotations_full) ? 128u : 96u);
				}

				if (!wanted)
					continue;

				if (options->per_track_rounding)
				{
					/* track_writer::get_rounding_policy (core/track_writer.h:97): per track only when seeking with per_track */
					policy = rounding_policy;
					if (rounding_policy == ACLO_ROUND_PER_TRACK)
						policy = options->track_rounding != NULL ? options->track_rounding[track_index] : ACLO_ROUND_NONE;

					/* decompress_track_v0 folds the policy into alpha and always interpolates (decompression.transform.h:1975-1983) */
					if (single)
					{
						alpha = aclo_apply_rounding_policy(alpha, policy);
						policy = ACLO_ROUND_NONE;
					}
				}

				if (kind == 0)
				{
					float q0[4], q1[4], result[4];
					unpack_rotation_sample(&ctx, 0, index, sample_bit_offsets[0], q0);
					unpack_rotation_sample(&ctx, 1, index, sample_bit_offsets[1], q1);
					/* :1416-1475: W is reconstructed (and, under 'always', the samples normalized) for the drop-W formats only; quatf_full
					 * samples carry their W. should_interpolate_samples (decompression_context.transform.h:192-199) is true whenever the
					 * settings support more than one rotation format -- the only settings that accept a full format besides a raw-only one --
					 * so the samples are always interpolated here. */
					if (!rotations_full)
					{
						q0[3] = quat_from_positive_w(q0[0], q0[1], q0[2]);
						q1[3] = quat_from_positive_w(q1[0], q1[1], q1[2]);

						/* animated_track_cache.transform.h:1463-1473 (whole pose only) */
						if (!single && options->normalization == ACLO_NORMALIZE_ALWAYS && options->per_track_rounding)
						{
							quat_normalize(q0);
							quat_normalize(q1);
						}
					}

					if (policy == ACLO_ROUND_FLOOR)
						memcpy(result, q0, sizeof(result));
					else if (policy == ACLO_ROUND_CEIL)
						memcpy(result, q1, sizeof(result));
					else if (policy == ACLO_ROUND_NEAREST)
						memcpy(result, seek.interpolation_alpha < 0.5f ? q0 : q1, sizeof(result));
					else
					{
						/* :1604-1616 */
						quat_lerp_no_normalization(q0, q1, alpha, result);
						if (options->normalization >= ACLO_NORMALIZE_LERP_ONLY)
							quat_normalize(result);
					}
					qvv[0] = result[0]; qvv[1] = result[1]; qvv[2] = result[2]; qvv[3] = result[3];
				}
				else
				{
					const uint8_t* clip_range_base = kind == 1 ? clip_range_translations : clip_range_scales;
					const uint32_t range_entries_before = kind == 1 ? num_rotations_padded : num_rotations_padded + num_translation_entries;
					float v0[3], v1[3], result[3];
					unpack_vector3_sample(&ctx, 0, format_index, seek.segment_range_data[0] + (uint64_t)range_entries_before * 6, clip_range_base, index, sample_bit_offsets[0], v0);
					unpack_vector3_sample(&ctx, 1, format_index, seek.segment_range_data[1] + (uint64_t)range_entries_before * 6, clip_range_base, index, sample_bit_offsets[1], v1);

					/* unpack_translation_group (:1774-1836) */
					if (policy == ACLO_ROUND_FLOOR)
						memcpy(result, v0, sizeof(result));
					else if (policy == ACLO_ROUND_CEIL)
						memcpy(result, v1, sizeof(result));
					else if (policy == ACLO_ROUND_NEAREST)
						memcpy(result, seek.interpolation_alpha < 0.5f ? v0 : v1, sizeof(result));
					else
						vector_lerp3(v0, v1, alpha, result);

					qvv[kind * 4 + 0] = result[0];
					qvv[kind * 4 + 1] = result[1];
					qvv[kind * 4 + 2] = result[2];
					qvv[kind * 4 + 3] = 0.0f;
				}
			}
		}
	}

	return 0;
}

int aclo_decompress_tracks(const void* blob, float sample_time, int rounding_policy, const aclo_options* options, float* out_pose)
{
	aclo_options defaults;
	if (options == NULL) { aclo_default_options(&defaults); options = &defaults; }
	return decode_pose(blob, sample_time, rounding_policy, options, -1, out_pose);
}

int aclo_decompress_track(const void* blob, float sample_time, int rounding_policy, const aclo_options* options, uint32_t track_index, float* out_qvv)
{
	aclo_options defaults;
	if (options == NULL) { aclo_default_options(&defaults); options = &defaults; }
	if ((int32_t)track_index < 0)
		return 2;
	return decode_pose(blob, sample_time, rounding_policy, options, (int)track_index, out_qvv);
}

int aclo_decompress_tracks_batch(const void* const* blobs, const uint32_t* clip_indices, const float* sample_times, uint32_t count,
	int rounding_policy, const aclo_options* options, float* out, uint64_t pose_stride_floats)
{
	uint32_t i;
	for (i = 0; i < count; ++i)
	{
		const int result = aclo_decompress_tracks(blobs[clip_indices[i]], sample_times[i], rounding_policy, options, out + (uint64_t)i * pose_stride_floats);
		if (result != 0)
			return result;
	}
	return 0;
}

/* ------------------------------------------------------------------------------------------------
 * Restatement of the reference's exhaustive packing unit test (tests/sources/math/test_vector4_packing.cpp:385-465,
 * unsigned half): every value of every bit width in [first_num_bits, last_num_bits] goes through
 * pack_vector3_uXX -> memcpy_bits at bit offsets {0,1,5,31,32,33,63,64,65,93} -> unpack_vector3_uXX and must come
 * back within 1e-6. Returns the number of mismatches (0 = pass).
 * ---------------------------------------------------------------------------------------------- */
/* ---- scalar tracks (float1f .. float4f, vector4f) ---------------------------------------------------------------------
 * scalar_tracks_header (core/impl/compressed_headers.h:140-165) follows the tracks_header; per track one bit rate byte, then
 * constant values, range values (min[C] extent[C] per quantized track) and the animated bitstream, frame major. */
typedef struct
{
	uint32_t num_bits_per_frame;
	uint32_t metadata_per_track, track_constant_values, track_range_values, track_animated_values;		/* offsets from this header */
} scalar_tracks_header_t;

/* core/impl/variable_bit_rates.h:42-45 */
static const uint8_t k_bit_rate_num_bits_v0[] = { 0, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 32 };
static const uint8_t k_bit_rate_num_bits[] = { 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22, 23, 32 };

/* get_track_num_sample_elements (core/track_types.h) */
uint32_t aclo_scalar_num_components(const void* blob)
{
	switch (get_tracks_header(blob)->track_type)
	{
	case 0: return 1;
	case 1: return 2;
	case 2: return 3;
	case 3: case 4: return 4;
	default: return 0;		/* qvvf and anything else: not a scalar track list */
	}
}

/* One component of unpack_scalarf_uXX / vector2_uXX / vector3_uXX / vector4_uXX (math/scalar_packing.h:113-160, math/vector4_packing.h:262-330,
 * 921-1035,1061-1130): 32 bits loaded at the byte holding the first bit, big endian, shifted, masked, converted, scaled. */
static float unpack_component_uXX(uint32_t num_bits, const uint8_t* data, uint32_t bit_offset)
{
	const uint32_t mask = (1u << num_bits) - 1u;
	const float inv_max_value = 1.0f / (float)mask;
	const uint32_t window = bswap32(load_u32(data + (bit_offset / 8)));
	const uint32_t value = (window >> ((32 - num_bits) - (bit_offset % 8))) & mask;
	return (float)(int32_t)value * inv_max_value;
}

/* One component of unpack_scalarf_32 / vector2_64 / vector3_96 / vector4_128 (math/scalar_packing.h:71-110, vector4_packing.h:59-164,400-470,479-599) */
static float unpack_component_32(const uint8_t* data, uint32_t bit_offset)
{
	uint64_t window = bswap64(load_u64(data + (bit_offset / 8)));
	window <<= bit_offset % 8;
	return bits_to_float((uint32_t)(window >> 32));
}

/* seek_v0 + decompress_tracks_v0 / decompress_track_v0 of decompression/impl/decompression.scalar.h:182-240,242-480,482-715.
 * track_filter < 0: every track. out: [num_tracks][num_components] floats (only the filtered track's slot is written otherwise). */
static int decode_scalar_tracks(const void* blob, float sample_time, int rounding_policy, const aclo_options* options, int track_filter, float* out)
{
	const tracks_header_t* header = get_tracks_header(blob);
	const uint32_t num_components = aclo_scalar_num_components(blob);
	if (num_components == 0)
		return 1;
	if (header->num_tracks == 0 || header->num_samples == 0)
		return 0;		/* empty track list */
	if (track_filter >= 0 && (uint32_t)track_filter >= header->num_tracks)
		return 0;		/* invalid track index: silently ignored (:496-498) */

	{
		const scalar_tracks_header_t* sh = (const scalar_tracks_header_t*)((const uint8_t*)blob + 32);
		const uint8_t* base = (const uint8_t*)sh;
		const uint8_t* metadata = base + sh->metadata_per_track;
		const uint8_t* constant_values = base + sh->track_constant_values;
		const uint8_t* range_values = base + sh->track_range_values;
		const uint8_t* animated_values = base + sh->track_animated_values;
		const uint8_t* num_bits_at_bit_rate = header->version == 7 ? k_bit_rate_num_bits_v0 : k_bit_rate_num_bits;

		const int looping_policy = resolve_looping_policy(header, options->looping_policy);
		const float duration = aclo_finite_duration(blob, looping_policy);
		uint32_t key_frame0, key_frame1, track_index, c;
		uint32_t track_bit_offset = 0;
		float interpolation_alpha;

		/* :189-190 scalar_clamp */
		sample_time = sample_time < 0.0f ? 0.0f : (sample_time > duration ? duration : sample_time);
		find_samples_with_rate(header->num_samples, header->sample_rate, sample_time, rounding_policy, looping_policy, &key_frame0, &key_frame1, &interpolation_alpha);

		for (track_index = 0; track_index < header->num_tracks; ++track_index)
		{
			const uint32_t num_bits = num_bits_at_bit_rate[metadata[track_index]];
			const int wanted = track_filter < 0 || (uint32_t)track_filter == track_index;
			float* value = out + (size_t)track_index * num_components;

			if (num_bits == 0)
			{
				/* constant bit rate: the sample lives in the constant values (:279-283) */
				if (wanted)
					for (c = 0; c < num_components; ++c)
						value[c] = load_f32(constant_values + 4 * c);
				constant_values += 4 * num_components;
				continue;
			}

			if (wanted)
			{
				float alpha = interpolation_alpha;
				if (options->per_track_rounding)
				{
					/* track_writer::get_rounding_policy, then apply_rounding_policy on the alpha the seek left behind (:246-258,273-279) */
					int policy = rounding_policy;
					if (rounding_policy == ACLO_ROUND_PER_TRACK)
						policy = options->track_rounding != NULL ? options->track_rounding[track_index] : ACLO_ROUND_NONE;
					alpha = aclo_apply_rounding_policy(interpolation_alpha, policy);
				}

				for (c = 0; c < num_components; ++c)
				{
					const uint32_t bit_offset0 = key_frame0 * sh->num_bits_per_frame + track_bit_offset + c * num_bits;
					const uint32_t bit_offset1 = key_frame1 * sh->num_bits_per_frame + track_bit_offset + c * num_bits;
					float value0, value1;
					if (num_bits == 32)
					{
						value0 = unpack_component_32(animated_values, bit_offset0);
						value1 = unpack_component_32(animated_values, bit_offset1);
					}
					else
					{
						const float range_min = load_f32(range_values + 4 * c);
						const float range_extent = load_f32(range_values + 4 * (num_components + c));
						value0 = unpack_component_uXX(num_bits, animated_values, bit_offset0);
						value1 = unpack_component_uXX(num_bits, animated_values, bit_offset1);
						value0 = (value0 * range_extent) + range_min;		/* scalar/vector_mul_add: multiply, then add */
						value1 = (value1 * range_extent) + range_min;
					}
					/* rtm::scalar_lerp / vector_lerp: (end * alpha) + (start - (start * alpha)) */
					value[c] = (value1 * alpha) + (value0 - (value0 * alpha));
				}
			}

			if (num_bits < 32)
				range_values += 8 * num_components;
			track_bit_offset += num_bits * num_components;
		}
	}
	return 0;
}

int aclo_scalar_decompress_tracks(const void* blob, float sample_time, int rounding_policy, const aclo_options* options, float* out)
{
	return decode_scalar_tracks(blob, sample_time, rounding_policy, options, -1, out);
}

int aclo_scalar_decompress_track(const void* blob, float sample_time, int rounding_policy, const aclo_options* options, uint32_t track_index, float* out_value)
{
	/* decode into the track's own slot of a virtual array: shift the base so that slot 'track_index' is out_value */
	const uint32_t num_components = aclo_scalar_num_components(blob);
	if (num_components == 0)
		return 1;
	return decode_scalar_tracks(blob, sample_time, rounding_policy, options, (int)track_index, out_value - (size_t)track_index * num_components);
}

/* decompress_tracks of a scalar track list for every instance of a batch (full-size GPU batches are compared instance by instance);
 * instance i's values go to out + i * row_stride_floats, num_tracks * C floats */
int aclo_scalar_decompress_tracks_batch(const void* const* blobs, const uint32_t* clip_indices, const float* sample_times, uint32_t count,
	int rounding_policy, const aclo_options* options, float* out, uint64_t row_stride_floats)
{
	uint32_t i;
	for (i = 0; i < count; ++i)
	{
		const int result = aclo_scalar_decompress_tracks(blobs[clip_indices[i]], sample_times[i], rounding_policy, options, out + (uint64_t)i * row_stride_floats);
		if (result != 0)
			return result;
	}
	return 0;
}

uint32_t aclo_selftest_pack_vector3_uXX(uint32_t first_num_bits, uint32_t last_num_bits)
{
	static const uint32_t offsets[] = { 0, 1, 5, 31, 32, 33, 63, 64, 65, 93 };
	uint32_t num_errors = 0;
	uint32_t num_bits;

	for (num_bits = first_num_bits; num_bits <= last_num_bits; ++num_bits)
	{
		const uint32_t max_value = (1u << num_bits) - 1;
		const float inv_max = 1.0f / (float)max_value;
		uint32_t value;

		for (value = 0; value <= max_value; value += 3)
		{
			uint8_t buffer[64];
			uint8_t shifted[64];
			float in[3], out[3];
			uint32_t i, c;
			const uint32_t v1 = value + 1 < max_value ? value + 1 : max_value;
			const uint32_t v2 = value + 2 < max_value ? value + 2 : max_value;

			/* unpack_scalar_unsigned (math/scalar_packing.h:50-56) clamped to [0, 1] */
			in[0] = (float)value * inv_max;
			in[1] = (float)v1 * inv_max;
			in[2] = (float)v2 * inv_max;
			for (c = 0; c < 3; ++c)
				in[c] = in[c] < 0.0f ? 0.0f : (in[c] > 1.0f ? 1.0f : in[c]);

			memset(buffer, 0, sizeof(buffer));
			aclo_pack_vector3_uXX(in, num_bits, buffer);
			aclo_unpack_vector3_uXX(num_bits, buffer, 0, out);
			for (c = 0; c < 3; ++c)
				if (!(fabsf(in[c] - out[c]) < 1.0e-6f))
					num_errors++;

			for (i = 0; i < sizeof(offsets) / sizeof(offsets[0]); ++i)
			{
				memset(shifted, 0, sizeof(shifted));
				aclo_memcpy_bits(shifted, offsets[i], buffer, 0, (uint64_t)num_bits * 3);
				aclo_unpack_vector3_uXX(num_bits, shifted, offsets[i], out);
				for (c = 0; c < 3; ++c)
					if (!(fabsf(in[c] - out[c]) < 1.0e-6f))
						num_errors++;
			}
		}
	}

	return num_errors;
}

/* ---- pose consumers: what callers do with a decompressed local pose -------------------------------------------
 * core/additive_utils.h:128-160 (apply_additive_to_base, transform_add0 / transform_add1) and
 * compression/transform_pose_utils.h:35-50 (local_to_object_space), on poses of 12 floats per transform
 * (rotation xyzw | translation xyz0 | scale xyz0).
 *
 * Both are written in Realtime Math (rtm::quat_mul, rtm::qvv_mul, rtm::qvv_normalize), an un-vendored git submodule
 * (external/rtm, absent from the checkout). Restated from RTM 2.x's published x86 (SSE2) forms:
 *   - quat_mul: per lane (a*rw + b*rx) + (c*ry + d*rz), signs folded into the products;
 *   - quat_mul_vector3(v, q) = quat_mul(quat_mul(conjugate(q), (v.xyz, 0)), q);
 *   - qvv_mul(lhs, rhs): rotation = quat_mul(lhs.r, rhs.r); translation = quat_mul_vector3(lhs.t * rhs.s, rhs.r) + rhs.t;
 *     scale = lhs.s * rhs.s. NEGATIVE scales go through 3x4 matrices instead (qvv_mul_through_matrices below);
 *   - qvv_normalize normalizes the rotation. RTM's x86 quat_normalize starts from the hardware reciprocal square root ESTIMATE
 *     (not reproducible between CPU vendors); restated with the reference's own deterministic normalize (quat_normalize above,
 *     acl/math/quatf.h:200-222 arithmetic). Agreement with an x86 build of the reference is therefore to a few ulp per level of
 *     the hierarchy, not bit for bit; tests/test_pose_consumers_oracle.py states the tolerance.
 * The W lanes of translation and scale are unspecified in the reference (numeric residue of the rotation); 0 here. */
void aclo_quat_mul(const float lhs[4], const float rhs[4], float out[4])
{
	const float lx = lhs[0], ly = lhs[1], lz = lhs[2], lw = lhs[3];
	const float rx = rhs[0], ry = rhs[1], rz = rhs[2], rw = rhs[3];
	const float x = ((rw * lx) + (rx * lw)) + ((ry * lz) + -(rz * ly));
	const float y = ((rw * ly) + -(rx * lz)) + ((ry * lw) + (rz * lx));
	const float z = ((rw * lz) + (rx * ly)) + (-(ry * lx) + (rz * lw));
	const float w = ((rw * lw) + -(rx * lx)) + (-(ry * ly) + -(rz * lz));
	out[0] = x; out[1] = y; out[2] = z; out[3] = w;
}

static void quat_mul_vector3(const float vector[3], const float rotation[4], float out[3])
{
	const float vector_quat[4] = { vector[0], vector[1], vector[2], 0.0f };
	const float inv_rotation[4] = { -rotation[0], -rotation[1], -rotation[2], rotation[3] };
	float tmp[4], result[4];
	aclo_quat_mul(inv_rotation, vector_quat, tmp);
	aclo_quat_mul(tmp, rotation, result);
	out[0] = result[0]; out[1] = result[1]; out[2] = result[2];
}

/* rtm::qvv_mul's route for NEGATIVE scales (mirrored rigs): when any scale component of either operand is negative a quaternion
 * cannot carry the reflection, and RTM 2.x composes 3x4 matrices instead --
 *   matrix_from_qvv(lhs) * matrix_from_qvv(rhs)            (row vectors: lhs first; every row ((x * X + y * Y) + z * Z) [+ W])
 *   matrix_remove_scale: every axis normalized by its own length (left alone below a squared length of 1e-8)
 *   each axis multiplied by the sign (+1 / -1, +1 for zero) of the matching component of scale = lhs.scale * rhs.scale
 *   rotation = quat_from_matrix (trace > 0, or the largest diagonal element), normalized; translation = the product's W row.
 * Restated from RTM's documented algorithm (rtm/qvvf.h qvv_mul, rtm/matrix3x4f.h, rtm/quatf.h quat_from_matrix); like everywhere
 * else in this file the reciprocal square roots are the correctly rounded 1 / sqrt (RTM's x86 form refines the RSQRTSS estimate).
 * Pinned independently of any reading of RTM by tests/test_pose_consumers_oracle.py: object-space transforms of mirrored
 * hierarchies against an fp64 matrix chain of the same local transforms. */
static int qvv_mul_takes_matrix_path(const float lhs[12], const float rhs[12])
{
	/* vector_any_less_than3(vector_min(lhs.scale, rhs.scale), zero) */
	uint32_t c;
	for (c = 0; c < 3; ++c)
		if ((lhs[8 + c] < rhs[8 + c] ? lhs[8 + c] : rhs[8 + c]) < 0.0f)
			return 1;
	return 0;
}

static void matrix_from_qvv(const float t[12], float m[4][3])
{
	const float x = t[0], y = t[1], z = t[2], w = t[3];
	const float x2 = x + x, y2 = y + y, z2 = z + z;
	const float xx = x * x2, xy = x * y2, xz = x * z2, yy = y * y2, yz = y * z2, zz = z * z2, wx = w * x2, wy = w * y2, wz = w * z2;
	m[0][0] = (1.0f - (yy + zz)) * t[8]; m[0][1] = (xy + wz) * t[8]; m[0][2] = (xz - wy) * t[8];
	m[1][0] = (xy - wz) * t[9]; m[1][1] = (1.0f - (xx + zz)) * t[9]; m[1][2] = (yz + wx) * t[9];
	m[2][0] = (xz + wy) * t[10]; m[2][1] = (yz - wx) * t[10]; m[2][2] = (1.0f - (xx + yy)) * t[10];
	m[3][0] = t[4]; m[3][1] = t[5]; m[3][2] = t[6];
}

static void quat_from_matrix(float m[4][3], float out[4])
{
	const float trace = (m[0][0] + m[1][1]) + m[2][2];
	if (trace > 0.0f)
	{
		const float inv_trace = 1.0f / sqrtf(trace + 1.0f);
		const float half_inv_trace = inv_trace * 0.5f;
		out[0] = (m[1][2] - m[2][1]) * half_inv_trace;
		out[1] = (m[2][0] - m[0][2]) * half_inv_trace;
		out[2] = (m[0][1] - m[1][0]) * half_inv_trace;
		out[3] = (1.0f / inv_trace) * 0.5f;
	}
	else
	{
		uint32_t best = 0, next, last;
		float pseudo_trace, inv_pseudo_trace, half_inv_pseudo_trace;
		if (m[1][1] > m[0][0])
			best = 1;
		if (m[2][2] > m[best][best])
			best = 2;
		next = (best + 1) % 3;
		last = (next + 1) % 3;
		pseudo_trace = ((1.0f + m[best][best]) - m[next][next]) - m[last][last];
		inv_pseudo_trace = 1.0f / sqrtf(pseudo_trace);
		half_inv_pseudo_trace = inv_pseudo_trace * 0.5f;
		out[best] = (1.0f / inv_pseudo_trace) * 0.5f;
		out[next] = half_inv_pseudo_trace * (m[best][next] + m[next][best]);
		out[last] = half_inv_pseudo_trace * (m[best][last] + m[last][best]);
		out[3] = half_inv_pseudo_trace * (m[next][last] - m[last][next]);
	}
	quat_normalize(out);
}

static void qvv_mul_through_matrices(const float lhs[12], const float rhs[12], float out[12])
{
	float l[4][3], r[4][3], product[4][3];
	uint32_t row, c;
	matrix_from_qvv(lhs, l);
	matrix_from_qvv(rhs, r);
	for (row = 0; row < 4; ++row)
		for (c = 0; c < 3; ++c)
		{
			float value = ((l[row][0] * r[0][c]) + (l[row][1] * r[1][c])) + (l[row][2] * r[2][c]);
			if (row == 3)
				value = r[3][c] + value;
			product[row][c] = value;
		}
	for (row = 0; row < 3; ++row)
	{
		const float scale = lhs[8 + row] * rhs[8 + row];
		const float sign = scale >= 0.0f ? 1.0f : -1.0f;
		const float length_squared = ((product[row][0] * product[row][0]) + (product[row][1] * product[row][1])) + (product[row][2] * product[row][2]);
		if (length_squared >= 1.0e-8f)
		{
			const float inv_length = 1.0f / sqrtf(length_squared);
			for (c = 0; c < 3; ++c)
				product[row][c] = product[row][c] * inv_length;
		}
		for (c = 0; c < 3; ++c)
			product[row][c] = product[row][c] * sign;
		out[8 + row] = scale;
	}
	quat_from_matrix(product, out + 0);
	out[4] = product[3][0]; out[5] = product[3][1]; out[6] = product[3][2];
	out[7] = 0.0f;
	out[11] = 0.0f;
}

void aclo_qvv_mul(const float lhs[12], const float rhs[12], float out[12])
{
	float rotation[4], scaled[3], rotated[3];
	uint32_t c;
	if (qvv_mul_takes_matrix_path(lhs, rhs))
	{
		float result[12];
		qvv_mul_through_matrices(lhs, rhs, result);
		memcpy(out, result, sizeof(result));
		return;
	}
	aclo_quat_mul(lhs + 0, rhs + 0, rotation);
	for (c = 0; c < 3; ++c)
		scaled[c] = lhs[4 + c] * rhs[8 + c];
	quat_mul_vector3(scaled, rhs + 0, rotated);
	for (c = 0; c < 3; ++c)
	{
		const float translation = rotated[c] + rhs[4 + c];
		const float scale = lhs[8 + c] * rhs[8 + c];
		out[4 + c] = translation;
		out[8 + c] = scale;
	}
	out[0] = rotation[0]; out[1] = rotation[1]; out[2] = rotation[2]; out[3] = rotation[3];
	out[7] = 0.0f;
	out[11] = 0.0f;
}

/* apply_additive_to_base (core/additive_utils.h:150-160): format 0 none (the additive pose passes through), 1 relative
 * (qvv_mul(additive, base)), 2 additive0, 3 additive1. Poses may alias. */
void aclo_apply_additive_to_base(int additive_format, const float* base_pose, const float* additive_pose, uint32_t num_transforms, float* out_pose)
{
	uint32_t i, c;
	for (i = 0; i < num_transforms; ++i)
	{
		const float* base = base_pose + (uint64_t)i * 12;
		const float* additive = additive_pose + (uint64_t)i * 12;
		float result[12];
		if (additive_format == 1)
			aclo_qvv_mul(additive, base, result);
		else if (additive_format == 2 || additive_format == 3)
		{
			/* transform_add0 / transform_add1 (:128-142) */
			aclo_quat_mul(additive + 0, base + 0, result + 0);
			for (c = 0; c < 3; ++c)
			{
				result[4 + c] = additive[4 + c] + base[4 + c];
				result[8 + c] = additive_format == 2 ? additive[8 + c] * base[8 + c] : (1.0f + additive[8 + c]) * base[8 + c];
			}
			result[7] = 0.0f;
			result[11] = 0.0f;
		}
		else
		{
			memcpy(result, additive, sizeof(result));
			result[7] = 0.0f;
			result[11] = 0.0f;
		}
		memcpy(out_pose + (uint64_t)i * 12, result, sizeof(result));
	}
}

/* local_to_object_space (compression/transform_pose_utils.h:35-50): transform 0 is the root, every other transform follows
 * its parent (parent_indices[i] < i; parent_indices[0] is not read). Extension used by the GPU path and mirrored here: a
 * parent of 0xFFFFFFFF marks a further root. Poses may alias. */
void aclo_local_to_object_space(const uint32_t* parent_indices, const float* local_pose, uint32_t num_transforms, float* out_object_pose)
{
	uint32_t i;
	for (i = 0; i < num_transforms; ++i)
	{
		float result[12];
		if (i == 0 || parent_indices[i] == 0xFFFFFFFFu)
			memcpy(result, local_pose + (uint64_t)i * 12, sizeof(result));
		else
		{
			aclo_qvv_mul(local_pose + (uint64_t)i * 12, out_object_pose + (uint64_t)parent_indices[i] * 12, result);
			quat_normalize(result);
		}
		result[7] = 0.0f;
		result[11] = 0.0f;
		memcpy(out_object_pose + (uint64_t)i * 12, result, sizeof(result));
	}
}

/* The consumers' pipeline for a batch, so that full-size GPU batches are compared instance by instance (the reference validates every
 * sample: tools/acl_compressor/sources/validate_tracks.cpp:92-260): decompress_tracks of (blobs[clip_indices[i]], sample_times[i]);
 * with an additive format, decompress_tracks of its base (blobs[base_clip_indices[i]], base_sample_times[i]) and
 * apply_additive_to_base; with parent_indices, local_to_object_space. All clips of one call have num_transforms tracks. */
int aclo_decompress_poses_batch(const void* const* blobs, const uint32_t* clip_indices, const float* sample_times, uint32_t count,
	int rounding_policy, const aclo_options* options, int additive_format, const uint32_t* base_clip_indices, const float* base_sample_times,
	const uint32_t* parent_indices, uint32_t num_transforms, float* out, uint64_t pose_stride_floats)
{
	uint32_t i;
	float* base_pose = additive_format != 0 ? (float*)malloc((size_t)num_transforms * 12 * sizeof(float)) : NULL;
	int result = 0;
	if (additive_format != 0 && base_pose == NULL)
		return -1;
	for (i = 0; i < count && result == 0; ++i)
	{
		float* pose = out + (uint64_t)i * pose_stride_floats;
		if (aclo_num_tracks(blobs[clip_indices[i]]) != num_transforms)
			result = -2;
		if (result == 0)
			result = aclo_decompress_tracks(blobs[clip_indices[i]], sample_times[i], rounding_policy, options, pose);
		if (result == 0 && additive_format != 0)
		{
			if (aclo_num_tracks(blobs[base_clip_indices[i]]) != num_transforms)
				result = -2;
			else
				result = aclo_decompress_tracks(blobs[base_clip_indices[i]], base_sample_times[i], rounding_policy, options, base_pose);
			if (result == 0)
				aclo_apply_additive_to_base(additive_format, base_pose, pose, num_transforms, pose);
		}
		if (result == 0 && parent_indices != NULL)
			aclo_local_to_object_space(parent_indices, pose, num_transforms, pose);
	}
	free(base_pose);
	return result;
}

/* Blend of K local poses ("blend of N clips"). The reference ships no such function -- only the arithmetic it is made
 * of: the sign bias and the normalize of its own quaternion interpolation (math/quatf.h:170-211). DEFINED here and in include/aclhip.h
 * (aclhip_pose_consumers::num_blend_clips), pinned by an fp64 restatement in tests/test_pose_consumers_oracle.py. Per transform, fp32,
 * one IEEE operation at a time, in this order:
 *   rotation     acc = q_0 * w_0
 *                for k = 1 .. K-1:  dot = ((acc.x q_k.x + acc.y q_k.y) + acc.z q_k.z) + acc.w q_k.w
 *                                   acc = (q_k * (dot < 0 ? -w_k : w_k)) + acc          (multiply, then add)
 *                rotation = acc * (1 / sqrt(((x x + y y) + z z) + w w))                  (quat_normalize above, math/quatf.h:200-211)
 *   translation  acc = t_0 * w_0;  acc = (t_k * w_k) + acc                               scale: like the translation
 * The weights are used as given (not normalized). poses[k]: 12 floats per transform. `out` may alias poses[0]. */
void aclo_blend_poses(const float* const* poses, const float* weights, uint32_t num_poses, uint32_t num_transforms, float* out)
{
	uint32_t i, k, c;
	for (i = 0; i < num_transforms; ++i)
	{
		const float* first = poses[0] + (uint64_t)i * 12;
		float result[12];
		for (c = 0; c < 4; ++c)
			result[c] = first[c] * weights[0];
		for (c = 0; c < 3; ++c)
		{
			result[4 + c] = first[4 + c] * weights[0];
			result[8 + c] = first[8 + c] * weights[0];
		}
		for (k = 1; k < num_poses; ++k)
		{
			const float* pose = poses[k] + (uint64_t)i * 12;
			float dot = result[0] * pose[0];
			float signed_weight;
			dot = dot + (result[1] * pose[1]);
			dot = dot + (result[2] * pose[2]);
			dot = dot + (result[3] * pose[3]);
			signed_weight = dot < 0.0f ? -weights[k] : weights[k];
			for (c = 0; c < 4; ++c)
				result[c] = (pose[c] * signed_weight) + result[c];
			for (c = 0; c < 3; ++c)
			{
				result[4 + c] = (pose[4 + c] * weights[k]) + result[4 + c];
				result[8 + c] = (pose[8 + c] * weights[k]) + result[8 + c];
			}
		}
		quat_normalize(result);
		result[7] = 0.0f;
		result[11] = 0.0f;
		memcpy(out + (uint64_t)i * 12, result, sizeof(result));
	}
}

/* aclo_decompress_poses_batch with a blend in front: instance i is the blend of num_blend_clips clip instances -- clip_indices[i] at
 * sample_times[i] first, then blend_clip_indices[i * (K - 1) + j] at blend_sample_times[i * (K - 1) + j] -- with weights
 * blend_weights[i * K + k]; the blended local pose then takes the additive apply and the object space conversion like a decoded one. */
int aclo_decompress_blended_poses_batch(const void* const* blobs, const uint32_t* clip_indices, const float* sample_times, uint32_t count,
	int rounding_policy, const aclo_options* options, uint32_t num_blend_clips, const uint32_t* blend_clip_indices, const float* blend_sample_times, const float* blend_weights,
	int additive_format, const uint32_t* base_clip_indices, const float* base_sample_times,
	const uint32_t* parent_indices, uint32_t num_transforms, float* out, uint64_t pose_stride_floats)
{
	uint32_t i, k;
	const size_t pose_floats = (size_t)num_transforms * 12;
	float* scratch;
	int result = 0;
	if (num_blend_clips < 2 || num_blend_clips > 8)
		return -3;
	scratch = (float*)malloc(pose_floats * sizeof(float) * (num_blend_clips + 1));
	if (scratch == NULL)
		return -1;
	for (i = 0; i < count && result == 0; ++i)
	{
		float* pose = out + (uint64_t)i * pose_stride_floats;
		const float* decoded[8];
		for (k = 0; k < num_blend_clips && result == 0; ++k)
		{
			const uint32_t clip = k == 0 ? clip_indices[i] : blend_clip_indices[(uint64_t)i * (num_blend_clips - 1) + (k - 1)];
			const float time = k == 0 ? sample_times[i] : blend_sample_times[(uint64_t)i * (num_blend_clips - 1) + (k - 1)];
			if (aclo_num_tracks(blobs[clip]) != num_transforms)
				result = -2;
			else
				result = aclo_decompress_tracks(blobs[clip], time, rounding_policy, options, scratch + pose_floats * k);
			decoded[k] = scratch + pose_floats * k;
		}
		if (result == 0)
			aclo_blend_poses(decoded, blend_weights + (uint64_t)i * num_blend_clips, num_blend_clips, num_transforms, pose);
		if (result == 0 && additive_format != 0)
		{
			float* base_pose = scratch + pose_floats * num_blend_clips;
			if (aclo_num_tracks(blobs[base_clip_indices[i]]) != num_transforms)
				result = -2;
			else
				result = aclo_decompress_tracks(blobs[base_clip_indices[i]], base_sample_times[i], rounding_policy, options, base_pose);
			if (result == 0)
				aclo_apply_additive_to_base(additive_format, base_pose, pose, num_transforms, pose);
		}
		if (result == 0 && parent_indices != NULL)
			aclo_local_to_object_space(parent_indices, pose, num_transforms, pose);
	}
	free(scratch);
	return result;
}

/* ---- small utilities the reference's unit tests pin (tests/sources/core/test_time_utils.cpp, test_bit_manip_utils.cpp,
 * tests/sources/math/test_scalar_packing.cpp): exposed so that tests/test_oracle_kats.py can restate those tests ---- */

/* core/impl/time_utils.impl.h:44-112 */
uint32_t aclo_calculate_num_samples(float duration, float sample_rate)
{
	if (duration == 0.0f)
		return 0;
	if (isinf(duration))
		return 1;
	return (uint32_t)floorf((duration * sample_rate) + 0.5f) + 1;
}

float aclo_calculate_duration(uint32_t num_samples, float sample_rate)
{
	if (num_samples == 0)
		return 0.0f;
	if (num_samples == 1)
		return INFINITY;
	return (float)(num_samples - 1) / sample_rate;
}

float aclo_calculate_finite_duration(uint32_t num_samples, float sample_rate)
{
	if (num_samples <= 1)
		return 0.0f;
	return (float)(num_samples - 1) / sample_rate;
}

/* core/bit_manip_utils.h (the forms the database seek relies on: 32 bit) */
uint32_t aclo_count_set_bits(uint32_t value) { return count_set_bits(value); }
uint32_t aclo_count_leading_zeros(uint32_t value) { return count_leading_zeros(value); }
uint32_t aclo_count_trailing_zeros(uint32_t value) { return count_trailing_zeros(value); }

/* math/scalar_packing.h:40-68. rtm::scalar_round_symmetric: half away from zero (floor(x + 0.5) for the non-negative inputs here) */
uint32_t aclo_pack_scalar_unsigned(float input, uint32_t num_bits)
{
	const uint32_t max_value = (1u << num_bits) - 1u;
	return (uint32_t)floorf(input * (float)max_value + 0.5f);
}

float aclo_unpack_scalar_unsigned(uint32_t input, uint32_t num_bits)
{
	const uint32_t max_value = (1u << num_bits) - 1u;
	const float inv_max_value = 1.0f / (float)max_value;
	return (float)input * inv_max_value;
}

uint32_t aclo_pack_scalar_signed(float input, uint32_t num_bits) { return aclo_pack_scalar_unsigned((input * 0.5f) + 0.5f, num_bits); }
float aclo_unpack_scalar_signed(uint32_t input, uint32_t num_bits) { return (aclo_unpack_scalar_unsigned(input, num_bits) * 2.0f) - 1.0f; }

/* The field readers of the scalar decode, at any bit offset (math/scalar_packing.h:71-160) */
float aclo_unpack_scalarf_32(const uint8_t* data, uint32_t bit_offset) { return unpack_component_32(data, bit_offset); }
float aclo_unpack_scalarf_uXX(uint32_t num_bits, const uint8_t* data, uint32_t bit_offset) { return unpack_component_uXX(num_bits, data, bit_offset); }

/* "scalar packing math" of test_scalar_packing.cpp:44-79 for num_bits in [first, last]: boundary values and the exhaustive
 * unpack -> pack round trip, unsigned and signed. Returns the number of violations. */
uint32_t aclo_selftest_scalar_packing(uint32_t first_num_bits, uint32_t last_num_bits)
{
	uint32_t num_errors = 0;
	uint32_t num_bits, value;
	for (num_bits = first_num_bits; num_bits <= last_num_bits; ++num_bits)
	{
		const uint32_t max_value = (1u << num_bits) - 1u;
		num_errors += aclo_pack_scalar_unsigned(0.0f, num_bits) != 0;
		num_errors += aclo_pack_scalar_unsigned(1.0f, num_bits) != max_value;
		num_errors += aclo_unpack_scalar_unsigned(0, num_bits) != 0.0f;
		num_errors += !(fabsf(aclo_unpack_scalar_unsigned(max_value, num_bits) - 1.0f) < 1.0e-6f);
		num_errors += aclo_pack_scalar_signed(-1.0f, num_bits) != 0;
		num_errors += aclo_pack_scalar_signed(1.0f, num_bits) != max_value;
		num_errors += aclo_unpack_scalar_signed(0, num_bits) != -1.0f;
		num_errors += !(fabsf(aclo_unpack_scalar_signed(max_value, num_bits) - 1.0f) < 1.0e-6f);
		for (value = 0; value < max_value; ++value)
		{
			const float unpacked0 = aclo_unpack_scalar_unsigned(value, num_bits);
			const float unpacked1 = aclo_unpack_scalar_signed(value, num_bits);
			if (aclo_pack_scalar_unsigned(unpacked0, num_bits) != value || unpacked0 < 0.0f || unpacked0 > 1.0f)
				num_errors++;
			if (aclo_pack_scalar_signed(unpacked1, num_bits) != value || unpacked1 < -1.0f || unpacked1 > 1.0f)
				num_errors++;
		}
	}
	return num_errors;
}
