// aclhip_device.h -- gfx950 device functions shared by the decode kernels: per-instance seek and the decode of
// one animated sub-track from the ACL bitstream. No host code here.
//
// Arithmetic follows the reference operation by operation (fp32, round to nearest, NO contraction -- this file
// must be compiled with -ffp-contract=off) so that poses match the reference CPU decoder:
//   seek_v0                         decompression/impl/decompression.transform.h:206-563
//   unpack_animated_quat            decompression/impl/animated_track_cache.transform.h:515-687
//   unpack_animated_vector3         decompression/impl/animated_track_cache.transform.h:871-990
//   remap_segment/clip_range_data4  decompression/impl/animated_track_cache.transform.h:302-350,391-466
//   quat_from_positive_w4 / quat_lerp_no_normalization4 / quat_normalize4   math/quatf.h:135-211
//   unpack_vector3_uXX / _96 / _u48 / _u24                                   math/vector4_packing.h:479-599,628-653,781-818,921-1035
// (paths relative to /root/reference/includes/acl)
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "acl_format.h"

namespace aclhip
{
	// Per clip record in HBM, written once at registration; read through the scalar cache by every wave.
	struct alignas(128) device_clip
	{
		const uint8_t* blob;				// the compressed_tracks bytes, unchanged, 16 byte aligned, >= 32 bytes of tail padding
		const float4* base_pose;			// [3 * num_tracks] rotation | translation | scale per track with default and constant sub-tracks expanded
		const uint32_t* quad_map;			// [3 * num_tracks] (class & 3) | (animated ordinal across rot,trans,scale) << 2
		const uint32_t* animated_tracks;	// [num animated sub-tracks] track index of every animated sub-track (rot, then trans, then scale)
		const uint8_t* db_headers;			// database runtime clip/segment headers (device) or null
		const uint8_t* db_bulk_data[2];		// database bulk data, medium / low importance tier (device) or null
		uint32_t num_tracks;
		uint32_t num_samples;
		float sample_rate;
		float duration_clamp;				// calculate_finite_duration(num_samples)
		float duration_wrap;				// calculate_finite_duration(num_samples + 1)
		uint32_t flags;						// k_clip_*
		uint32_t num_segments;
		uint32_t segment_headers_offset;	// from the blob start
		uint32_t segment_header_size;		// 16, or 20 with stripped keyframes / database
		uint32_t num_animated_rotations;
		uint32_t num_animated_translations;
		uint32_t num_animated_scales;
		uint32_t num_animated_variable;		// rotations padded to 4 + translations + scales
		uint32_t clip_range_offset;			// from the blob start
		uint32_t raw_num_bits;				// 31 from v02_01_99_1 on, 32 before
		uint32_t db_clip_header_offset;		// into db_headers
	};

	constexpr uint32_t k_clip_has_scale = 1u << 0;
	constexpr uint32_t k_clip_has_stripped_keyframes = 1u << 1;	// stripped keyframes or database: 20 byte segment headers
	constexpr uint32_t k_clip_has_database = 1u << 2;
	constexpr uint32_t k_clip_wraps = 1u << 3;					// compressed_tracks::get_looping_policy() == wrap
	constexpr uint32_t k_clip_valid = 1u << 31;

	// Launch wide settings (aclhip_decompress_params resolved to device pointers)
	struct decode_params
	{
		const float* default_values;
		const uint8_t* track_rounding_policies;
		const uint8_t* instance_rounding_policies;
		uint8_t rounding_policy;
		uint8_t looping_policy;
		uint8_t normalization;
		uint8_t per_track_rounding;
		uint8_t default_modes[3];
		uint8_t pad;
	};

	// What seek leaves behind for the decode (persistent_transform_decompression_context_v0, decompression_context.transform.h:53-116)
	struct seek_state
	{
		const uint8_t* format_per_track_data[2];
		const uint8_t* segment_range_data[2];
		const uint8_t* animated_track_data[2];
		uint32_t key_frame_bit_offsets[2];
		float interpolation_alpha;
		bool uses_single_segment;
		bool has_segments;
	};

	__device__ __forceinline__ uint32_t load_u32(const uint8_t* p) { return *reinterpret_cast<const uint32_t*>(p); }				// 4 byte aligned
	__device__ __forceinline__ uint64_t load_u64_aligned8(const uint8_t* p) { return *reinterpret_cast<const uint64_t*>(p); }
	__device__ __forceinline__ float load_f32(const uint8_t* p) { return *reinterpret_cast<const float*>(p); }						// 4 byte aligned
	__device__ __forceinline__ uint64_t load_u64_unaligned(const uint8_t* p) { uint64_t v; __builtin_memcpy(&v, p, 8); return v; }	// any alignment

	// core/impl/interpolation_utils.impl.h:261-278
	__device__ __forceinline__ float apply_rounding_policy(float alpha, uint32_t policy)
	{
		if (policy == k_round_floor) return 0.0f;
		if (policy == k_round_ceil) return 1.0f;
		if (policy == k_round_nearest) return floorf(alpha + 0.5f);
		return alpha;
	}

	// core/impl/interpolation_utils.impl.h:224-253 with rounding_policy == none
	__device__ __forceinline__ float find_linear_interpolation_alpha(float sample_index, uint32_t index0, uint32_t index1)
	{
		if (index0 == index1)
			return 0.0f;
		if (index0 < index1)
			return (sample_index - float(index0)) / float(index1 - index0);
		return sample_index - float(index0);
	}

	__device__ __forceinline__ void get_segment_data(const device_clip& clip, const uint8_t* segment_header,
		const uint8_t*& out_format, const uint8_t*& out_range, const uint8_t*& out_animated)
	{
		// core/impl/compressed_headers.h:309-324: offsets are relative to the transform_tracks_header, alignment is absolute
		const uint32_t segment_data = load_u32(segment_header + 12) + k_transform_header_offset;
		const uint32_t range_offset = align_to_u32(segment_data + clip.num_animated_variable, 2);
		const uint32_t range_size = clip.num_segments > 1 ? 6u * clip.num_animated_variable : 0u;
		const uint32_t animated_offset = align_to_u32(range_offset + range_size, 4);
		out_format = clip.blob + segment_data;
		out_range = clip.blob + range_offset;
		out_animated = clip.blob + animated_offset;
	}

	// seek_v0 (decompression/impl/decompression.transform.h:206-563). Everything here is wave uniform in the pose kernel.
	__device__ __forceinline__ void seek(const device_clip& clip, float sample_time, uint32_t rounding_policy, uint32_t looping_policy, seek_state& out)
	{
		const bool wrap = looping_policy == k_loop_as_compressed ? (clip.flags & k_clip_wraps) != 0 : looping_policy == k_loop_wrap;
		const float clip_duration = wrap ? clip.duration_wrap : clip.duration_clamp;
		const uint32_t num_samples = clip.num_samples;
		const uint32_t num_segments = clip.num_segments;
		const bool has_stripped_keyframes = (clip.flags & k_clip_has_stripped_keyframes) != 0;
		const bool has_database = (clip.flags & k_clip_has_database) != 0 && clip.db_headers != nullptr;
		const uint8_t* segment_headers = clip.blob + clip.segment_headers_offset;

		// :215-216 scalar_clamp
		sample_time = fminf(fmaxf(sample_time, 0.0f), clip_duration);

		// find_linear_interpolation_samples_with_sample_rate (core/impl/interpolation_utils.impl.h:143-201)
		const uint32_t last_sample_index = num_samples - 1;
		float sample_index = sample_time * clip.sample_rate;
		uint32_t key_frame0 = uint32_t(sample_index);
		uint32_t key_frame1;
		if (!wrap)
			key_frame1 = min(key_frame0 + 1, last_sample_index);
		else if (key_frame0 > last_sample_index)
		{
			sample_index = 0.0f;
			key_frame0 = 0;
			key_frame1 = 0;
		}
		else
			key_frame1 = key_frame0 + 1 >= num_samples ? 0 : key_frame0 + 1;

		float alpha = apply_rounding_policy(sample_index - float(key_frame0), rounding_policy);

		uint32_t segment_index0 = 0;
		uint32_t segment_index1 = 0;
		uint32_t segment_key_frame0 = key_frame0;
		uint32_t segment_key_frame1 = key_frame1;
		uint32_t segment_start0 = 0;
		uint32_t segment_start1 = 0;

		if (num_segments > 1)
		{
			// :372-409: guess, then scan at most 4 start indices (the list ends with a 0xFFFFFFFF sentinel)
			const uint8_t* segment_start_indices = clip.blob + (k_transform_header_offset + k_segment_start_indices_offset);
			const uint32_t approx_num_samples_per_segment = num_samples / num_segments;
			const uint32_t approx_segment_index = key_frame0 / approx_num_samples_per_segment;
			const uint32_t start_segment_index = approx_segment_index > 0 ? (approx_segment_index - 1) : 0;

			for (uint32_t i = 0; i < 4; ++i)
			{
				const uint32_t segment_index = start_segment_index + i;
				const uint32_t segment_start = load_u32(segment_start_indices + 4 * segment_index);
				if (key_frame0 < segment_start)
				{
					segment_index0 = segment_index - 1;
					if (key_frame1 == 0)
						segment_index1 = 0;		// wrapped around: first segment
					else
						segment_index1 = key_frame1 < segment_start ? segment_index0 : segment_index;
					break;
				}
			}

			segment_start0 = load_u32(segment_start_indices + 4 * segment_index0);
			segment_start1 = load_u32(segment_start_indices + 4 * segment_index1);
			segment_key_frame0 = key_frame0 - segment_start0;
			segment_key_frame1 = key_frame1 - segment_start1;
		}

		const uint8_t* segment_header0 = segment_headers + clip.segment_header_size * segment_index0;
		const uint8_t* segment_header1 = segment_headers + clip.segment_header_size * segment_index1;

		const uint8_t* db_animated_track_data0 = nullptr;
		const uint8_t* db_animated_track_data1 = nullptr;

		if (has_stripped_keyframes)
		{
			// :272-362 / :411-515: snap to the nearest keyframes that are present, in the clip or in a streamed database tier
			uint32_t sample_indices0 = load_u32(segment_header0 + 16);
			uint32_t sample_indices1 = load_u32(segment_header1 + 16);
			const float clip_sample_index = alpha + float(key_frame0);

			uint64_t medium0 = 0, medium1 = 0, low0 = 0, low1 = 0;
			if (has_database)
			{
				const uint8_t* db_segment_headers = clip.db_headers + clip.db_clip_header_offset + sizeof(database_runtime_clip_header);
				const uint64_t* tiers0 = reinterpret_cast<const uint64_t*>(db_segment_headers + sizeof(database_runtime_segment_header) * segment_index0);
				const uint64_t* tiers1 = reinterpret_cast<const uint64_t*>(db_segment_headers + sizeof(database_runtime_segment_header) * segment_index1);
				medium0 = __hip_atomic_load(tiers0 + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				low0 = __hip_atomic_load(tiers0 + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				medium1 = __hip_atomic_load(tiers1 + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				low1 = __hip_atomic_load(tiers1 + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				sample_indices0 |= uint32_t(medium0) | uint32_t(low0);
				sample_indices1 |= uint32_t(medium1) | uint32_t(low1);
			}

			const uint32_t candidate_indices0 = sample_indices0 & (0xFFFFFFFFu << (31 - segment_key_frame0));
			segment_key_frame0 = 31 - uint32_t(__builtin_ctz(candidate_indices0));
			const uint32_t candidate_indices1 = sample_indices1 & (0xFFFFFFFFu >> segment_key_frame1);
			segment_key_frame1 = uint32_t(__builtin_clz(candidate_indices1));

			alpha = find_linear_interpolation_alpha(clip_sample_index, segment_start0 + segment_key_frame0, segment_start1 + segment_key_frame1);

			sample_indices0 = load_u32(segment_header0 + 16);
			sample_indices1 = load_u32(segment_header1 + 16);

			if (has_database)
			{
				const uint64_t sample_bit0 = uint64_t(1) << (31 - segment_key_frame0);
				const uint64_t sample_bit1 = uint64_t(1) << (31 - segment_key_frame1);
				if ((medium0 & sample_bit0) != 0) { sample_indices0 = uint32_t(medium0); db_animated_track_data0 = clip.db_bulk_data[0] + uint32_t(medium0 >> 32); }
				else if ((low0 & sample_bit0) != 0) { sample_indices0 = uint32_t(low0); db_animated_track_data0 = clip.db_bulk_data[1] + uint32_t(low0 >> 32); }
				if ((medium1 & sample_bit1) != 0) { sample_indices1 = uint32_t(medium1); db_animated_track_data1 = clip.db_bulk_data[0] + uint32_t(medium1 >> 32); }
				else if ((low1 & sample_bit1) != 0) { sample_indices1 = uint32_t(low1); db_animated_track_data1 = clip.db_bulk_data[1] + uint32_t(low1 >> 32); }
			}

			// ordinal among the keyframes stored by the chosen data source
			segment_key_frame0 = uint32_t(__builtin_popcount(~(0xFFFFFFFFu >> segment_key_frame0) & sample_indices0));
			segment_key_frame1 = uint32_t(__builtin_popcount(~(0xFFFFFFFFu >> segment_key_frame1) & sample_indices1));
		}

		// :530-562
		get_segment_data(clip, segment_header0, out.format_per_track_data[0], out.segment_range_data[0], out.animated_track_data[0]);
		get_segment_data(clip, segment_header1, out.format_per_track_data[1], out.segment_range_data[1], out.animated_track_data[1]);
		if (db_animated_track_data0 != nullptr)
			out.animated_track_data[0] = db_animated_track_data0;
		if (db_animated_track_data1 != nullptr)
			out.animated_track_data[1] = db_animated_track_data1;

		out.key_frame_bit_offsets[0] = segment_key_frame0 * load_u32(segment_header0 + 0);
		out.key_frame_bit_offsets[1] = segment_key_frame1 * load_u32(segment_header1 + 0);
		out.interpolation_alpha = alpha;
		out.uses_single_segment = segment_index0 == segment_index1;
		out.has_segments = num_segments > 1;
	}

	// Bits a sub-track occupies per component in the animated pose (count_animated_group_bit_size, animated_track_cache.transform.h:1105-1192)
	__device__ __forceinline__ uint32_t stored_bits_per_component(uint32_t num_bits, uint32_t raw_num_bits) { return num_bits == raw_num_bits ? 32u : num_bits; }

	// Where one animated sub-track lives: kind, ordinal within kind, index of its format byte
	struct animated_slot
	{
		uint32_t kind;			// 0 rotation, 1 translation, 2 scale
		uint32_t index;			// ordinal within the kind
		uint32_t format_index;	// into format_per_track_data (rotations are padded to a multiple of 4)
	};

	__device__ __forceinline__ animated_slot make_animated_slot(const device_clip& clip, uint32_t animated_ordinal)
	{
		const uint32_t num_rotations = clip.num_animated_rotations;
		const uint32_t num_rotations_padded = (num_rotations + 3u) & ~3u;
		animated_slot slot;
		if (animated_ordinal < num_rotations)
		{
			slot.kind = 0;
			slot.index = animated_ordinal;
			slot.format_index = animated_ordinal;
		}
		else
		{
			const uint32_t vector_index = animated_ordinal - num_rotations;		// translations then scales share one AOS region
			slot.kind = vector_index < clip.num_animated_translations ? 1 : 2;
			slot.index = vector_index;
			slot.format_index = num_rotations_padded + vector_index;
		}
		return slot;
	}

	// The x, y, z of one sub-track at one keyframe, range expanded. `bit_offset` is relative to animated_track_data.
	__device__ __forceinline__ void unpack_animated_sample(const device_clip& clip, const seek_state& state, uint32_t key, const animated_slot& slot,
		uint32_t num_bits, uint32_t bit_offset, float out_xyz[3])
	{
		const bool is_rotation = slot.kind == 0;
		const uint32_t num_rotations = clip.num_animated_rotations;
		const uint32_t num_rotations_padded = (num_rotations + 3u) & ~3u;

		// Segment range bytes of this sub-track: rotations are SOA in groups of 4 (stride 4 between the six values),
		// translations/scales are AOS (stride 1)
		const uint8_t* segment_range = state.segment_range_data[key];
		uint32_t segment_range_stride;
		if (is_rotation)
		{
			segment_range += (slot.index >> 2) * 24u + (slot.index & 3u);
			segment_range_stride = 4;
		}
		else
		{
			segment_range += num_rotations_padded * 6u + slot.index * 6u;
			segment_range_stride = 1;
		}

		const bool is_constant_in_segment = num_bits == 0;
		const bool is_raw = num_bits == clip.raw_num_bits;
		const bool needs_segment_range = state.has_segments && !is_raw;		// width 0 reads its sample from the same bytes

		uint32_t range_bytes[6] = { 0, 0, 0, 0, 0, 0 };
		if (needs_segment_range)
		{
			#pragma unroll
			for (uint32_t i = 0; i < 6; ++i)
				range_bytes[i] = segment_range[i * segment_range_stride];
		}

		float xyz[3];
		if (is_constant_in_segment)
		{
			// animated_track_cache.transform.h:552-588 (rotation: hi/lo bytes split across SOA rows),
			// math/vector4_packing.h:628-653 (vector3: little endian u16)
			uint32_t x, y, z;
			if (is_rotation)
			{
				x = (range_bytes[0] << 8) | range_bytes[1];
				y = (range_bytes[2] << 8) | range_bytes[3];
				z = (range_bytes[4] << 8) | range_bytes[5];
			}
			else
			{
				x = (range_bytes[1] << 8) | range_bytes[0];
				y = (range_bytes[3] << 8) | range_bytes[2];
				z = (range_bytes[5] << 8) | range_bytes[4];
			}
			xyz[0] = float(x) * (1.0f / 65535.0f);
			xyz[1] = float(y) * (1.0f / 65535.0f);
			xyz[2] = float(z) * (1.0f / 65535.0f);
		}
		else if (is_raw)
		{
			// math/vector4_packing.h:479-599: three big endian floats at an arbitrary bit
			const uint8_t* data = state.animated_track_data[key] + (bit_offset >> 3);
			const uint32_t shift = bit_offset & 7u;
			#pragma unroll
			for (uint32_t c = 0; c < 3; ++c)
			{
				uint64_t window = __builtin_bswap64(load_u64_unaligned(data + 4 * c));
				window <<= shift;
				xyz[c] = __uint_as_float(uint32_t(window >> 32));
			}
		}
		else
		{
			// math/vector4_packing.h:921-1035. x and y come out of one 64 bit big endian window (7 + 2 * 23 <= 64), z out of a second one
			const uint32_t mask = (1u << num_bits) - 1u;
			const float inv_max_value = 1.0f / float(mask);
			const uint8_t* data = state.animated_track_data[key];

			const uint64_t window_xy = __builtin_bswap64(load_u64_unaligned(data + (bit_offset >> 3)));
			const uint32_t shift_xy = bit_offset & 7u;
			const uint32_t x = uint32_t(window_xy >> (64u - shift_xy - num_bits)) & mask;
			const uint32_t y = uint32_t(window_xy >> (64u - shift_xy - 2u * num_bits)) & mask;

			const uint32_t bit_offset_z = bit_offset + 2u * num_bits;
			const uint64_t window_z = __builtin_bswap64(load_u64_unaligned(data + (bit_offset_z >> 3)));
			const uint32_t z = uint32_t(window_z >> (64u - (bit_offset_z & 7u) - num_bits)) & mask;

			xyz[0] = float(x) * inv_max_value;
			xyz[1] = float(y) * inv_max_value;
			xyz[2] = float(z) * inv_max_value;
		}

		const bool ignore_segment_range = is_constant_in_segment || is_raw;
		const bool ignore_clip_range = is_raw;

		if (is_rotation)
		{
			// Whole pose flavour: ignored lanes still see a multiply by 1 and an add of 0
			// (remap_segment_range_data4 / remap_clip_range_data4, animated_track_cache.transform.h:316-349,420-465)
			if (state.has_segments)
			{
				#pragma unroll
				for (uint32_t c = 0; c < 3; ++c)
				{
					const float range_min = ignore_segment_range ? 0.0f : float(range_bytes[c]) * (1.0f / 255.0f);
					const float range_extent = ignore_segment_range ? 1.0f : float(range_bytes[3 + c]) * (1.0f / 255.0f);
					xyz[c] = (xyz[c] * range_extent) + range_min;
				}
			}

			const uint32_t group = slot.index >> 2;
			const uint32_t group_size = min(num_rotations - group * 4u, 4u);
			const uint8_t* clip_range = clip.blob + clip.clip_range_offset + group * 96u + (slot.index & 3u) * 4u;
			#pragma unroll
			for (uint32_t c = 0; c < 3; ++c)
			{
				const float range_min = ignore_clip_range ? 0.0f : load_f32(clip_range + group_size * 4u * c);
				const float range_extent = ignore_clip_range ? 1.0f : load_f32(clip_range + group_size * 4u * (3u + c));
				xyz[c] = (xyz[c] * range_extent) + range_min;
			}
		}
		else
		{
			// unpack_animated_vector3, animated_track_cache.transform.h:930-960
			if (state.has_segments && !ignore_segment_range)
			{
				#pragma unroll
				for (uint32_t c = 0; c < 3; ++c)
				{
					const float range_min = float(range_bytes[c]) * (1.0f / 255.0f);
					const float range_extent = float(range_bytes[3 + c]) * (1.0f / 255.0f);
					xyz[c] = (xyz[c] * range_extent) + range_min;
				}
			}

			if (!ignore_clip_range)
			{
				const uint8_t* clip_range = clip.blob + clip.clip_range_offset + num_rotations * 24u + slot.index * 24u;
				#pragma unroll
				for (uint32_t c = 0; c < 3; ++c)
					xyz[c] = (xyz[c] * load_f32(clip_range + 12u + 4u * c)) + load_f32(clip_range + 4u * c);
			}
		}

		out_xyz[0] = xyz[0];
		out_xyz[1] = xyz[1];
		out_xyz[2] = xyz[2];
	}

	// math/quatf.h:135-147
	__device__ __forceinline__ float quat_from_positive_w(float x, float y, float z)
	{
		float w_squared = 1.0f - (x * x);
		w_squared = w_squared - (y * y);
		w_squared = w_squared - (z * z);
		return sqrtf(fabsf(w_squared));
	}

	// math/quatf.h:200-211
	__device__ __forceinline__ float4 quat_normalize(float4 q)
	{
		float dot = q.x * q.x;
		dot = (q.y * q.y) + dot;
		dot = (q.z * q.z) + dot;
		dot = (q.w * q.w) + dot;
		const float inv_len = 1.0f / sqrtf(dot);
		return make_float4(q.x * inv_len, q.y * inv_len, q.z * inv_len, q.w * inv_len);
	}

	// math/quatf.h:170-196
	__device__ __forceinline__ float4 quat_lerp_no_normalization(float4 q0, float4 q1, float alpha)
	{
		float dot = q0.x * q1.x;
		dot = (q0.y * q1.y) + dot;
		dot = (q0.z * q1.z) + dot;
		dot = (q0.w * q1.w) + dot;
		const uint32_t bias = __float_as_uint(dot) & 0x80000000u;
		float4 result;
		result.x = (__uint_as_float(__float_as_uint(q1.x) ^ bias) * alpha) + (q0.x - (q0.x * alpha));
		result.y = (__uint_as_float(__float_as_uint(q1.y) ^ bias) * alpha) + (q0.y - (q0.y * alpha));
		result.z = (__uint_as_float(__float_as_uint(q1.z) ^ bias) * alpha) + (q0.z - (q0.z * alpha));
		result.w = (__uint_as_float(__float_as_uint(q1.w) ^ bias) * alpha) + (q0.w - (q0.w * alpha));
		return result;
	}

	// rtm::vector_lerp in its stable form: end * alpha + (start - start * alpha)
	__device__ __forceinline__ float lerp_stable(float start, float end, float alpha) { return (end * alpha) + (start - (start * alpha)); }

	// Decodes animated sub-track `slot` of one instance: both keyframes, range expansion, W reconstruction, interpolation.
	// `policy` is the effective rounding policy of the track (none unless per track rounding is enabled);
	// `lerp_alpha` the alpha handed to the interpolation.
	__device__ __forceinline__ float4 decode_animated_sub_track(const device_clip& clip, const seek_state& state, const animated_slot& slot,
		uint32_t num_bits0, uint32_t num_bits1, uint32_t bit_offset0, uint32_t bit_offset1,
		uint32_t policy, float lerp_alpha, uint32_t normalization, bool normalize_samples)
	{
		float v0[3], v1[3];
		unpack_animated_sample(clip, state, 0, slot, num_bits0, bit_offset0, v0);
		unpack_animated_sample(clip, state, 1, slot, num_bits1, bit_offset1, v1);

		if (slot.kind == 0)
		{
			float4 q0 = make_float4(v0[0], v0[1], v0[2], quat_from_positive_w(v0[0], v0[1], v0[2]));
			float4 q1 = make_float4(v1[0], v1[1], v1[2], quat_from_positive_w(v1[0], v1[1], v1[2]));

			// animated_track_cache.transform.h:1463-1473
			if (normalize_samples)
			{
				q0 = quat_normalize(q0);
				q1 = quat_normalize(q1);
			}

			if (policy == k_round_floor)
				return q0;
			if (policy == k_round_ceil)
				return q1;
			if (policy == k_round_nearest)
				return state.interpolation_alpha < 0.5f ? q0 : q1;

			// :1604-1616
			float4 result = quat_lerp_no_normalization(q0, q1, lerp_alpha);
			if (normalization >= 1)
				result = quat_normalize(result);
			return result;
		}

		// unpack_translation_group / unpack_scale_group, animated_track_cache.transform.h:1774-1836,1896-1958
		if (policy == k_round_floor)
			return make_float4(v0[0], v0[1], v0[2], 0.0f);
		if (policy == k_round_ceil)
			return make_float4(v1[0], v1[1], v1[2], 0.0f);
		if (policy == k_round_nearest)
			return state.interpolation_alpha < 0.5f ? make_float4(v0[0], v0[1], v0[2], 0.0f) : make_float4(v1[0], v1[1], v1[2], 0.0f);

		return make_float4(lerp_stable(v0[0], v1[0], lerp_alpha), lerp_stable(v0[1], v1[1], lerp_alpha), lerp_stable(v0[2], v1[2], lerp_alpha), 0.0f);
	}
}
