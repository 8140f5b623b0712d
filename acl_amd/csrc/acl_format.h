// acl_format.h -- the ACL `compressed_tracks` binary layout (qvvf transform tracks, uniformly sampled),
// restated for host C++ and gfx950 device code. The layout is kept byte-for-byte so that blobs produced
// by the reference compressor can be registered unchanged.
//
// Reference layout (file:line relative to /root/reference/includes/acl):
//   raw_buffer_header            core/impl/compressed_headers.h:51-58
//   tracks_header                core/impl/compressed_headers.h:61-131
//   segment_header               core/impl/compressed_headers.h:171-186
//   stripped_segment_header_t    core/impl/compressed_headers.h:193-197
//   packed_sub_track_types       core/impl/compressed_headers.h:214-224
//   transform_tracks_header      core/impl/compressed_headers.h:227-325
//   database runtime headers     core/impl/compressed_headers.h:397-439
//   versions                     core/compressed_tracks_version.h:44-88
//   formats                      core/track_formats.h:48-71
#pragma once

#include <stdint.h>

#if defined(__HIPCC__)
	#define ACLHIP_HD __host__ __device__ inline
#else
	#define ACLHIP_HD inline
#endif

namespace aclhip
{
	constexpr uint32_t k_tag_compressed_tracks = 0xac11ac11u;		// core/buffer_tag.h
	constexpr uint32_t k_tag_compressed_database = 0xac11db01u;
	constexpr uint32_t k_invalid_offset = 0xFFFFFFFFu;				// core/ptr_offset.h:123

	constexpr uint16_t k_version_first = 7;				// v02_00_00
	constexpr uint16_t k_version_v02_01_99 = 8;
	constexpr uint16_t k_version_v02_01_99_1 = 9;		// raw bit rate stored as 31 from here on
	constexpr uint16_t k_version_latest = 10;			// v02_01_00

	constexpr uint8_t k_algorithm_uniformly_sampled = 0;
	constexpr uint8_t k_track_type_float1f = 0;			// core/track_types.h:51-57: scalar track lists, 1 / 2 / 3 / 4 / 4 floats per sample
	constexpr uint8_t k_track_type_float2f = 1;
	constexpr uint8_t k_track_type_float3f = 2;
	constexpr uint8_t k_track_type_float4f = 3;
	constexpr uint8_t k_track_type_vector4f = 4;
	constexpr uint8_t k_track_type_qvvf = 12;

	constexpr uint8_t k_rotation_quatf_full = 0;
	constexpr uint8_t k_rotation_quatf_drop_w_full = 2;
	constexpr uint8_t k_rotation_quatf_drop_w_variable = 3;
	constexpr uint8_t k_vector_vector3f_full = 0;
	constexpr uint8_t k_vector_vector3f_variable = 1;

	// sample_rounding_policy (core/sample_rounding_policy.h)
	constexpr uint8_t k_round_none = 0;
	constexpr uint8_t k_round_floor = 1;
	constexpr uint8_t k_round_ceil = 2;
	constexpr uint8_t k_round_nearest = 3;
	constexpr uint8_t k_round_per_track = 4;

	// sample_looping_policy (core/sample_looping_policy.h)
	constexpr uint8_t k_loop_clamp = 0;
	constexpr uint8_t k_loop_wrap = 1;
	constexpr uint8_t k_loop_as_compressed = 2;

	// Sub-track classes, 2 bits each, 16 per u32, first track in the top bits
	constexpr uint32_t k_sub_track_default = 0;
	constexpr uint32_t k_sub_track_constant = 1;
	constexpr uint32_t k_sub_track_animated = 2;

	struct raw_buffer_header
	{
		uint32_t size;		// whole blob, bytes
		uint32_t hash;		// FNV-1a 32 over bytes [8, size)
	};

	struct tracks_header
	{
		uint32_t tag;
		uint16_t version;
		uint8_t algorithm_type;
		uint8_t track_type;
		uint32_t num_tracks;
		uint32_t num_samples;
		float sample_rate;
		uint32_t misc_packed;

		ACLHIP_HD bool has_scale() const { return (misc_packed & 1u) != 0; }
		ACLHIP_HD uint32_t default_scale() const { return (misc_packed >> 1) & 1u; }
		ACLHIP_HD uint8_t scale_format() const { return uint8_t((misc_packed >> 2) & 1u); }
		ACLHIP_HD uint8_t translation_format() const { return uint8_t((misc_packed >> 3) & 1u); }
		ACLHIP_HD uint8_t rotation_format() const { return uint8_t((misc_packed >> 4) & 15u); }
		ACLHIP_HD bool has_database() const { return (misc_packed & (1u << 8)) != 0; }
		ACLHIP_HD bool has_trivial_default_values() const { return (misc_packed & (1u << 9)) != 0; }
		ACLHIP_HD bool has_stripped_keyframes() const { return (misc_packed & (1u << 10)) != 0; }
		ACLHIP_HD bool is_wrap_optimized() const { return (misc_packed & (1u << 30)) != 0; }
		ACLHIP_HD bool has_metadata() const { return (misc_packed >> 31) != 0; }
	};

	struct segment_header
	{
		uint32_t animated_pose_bit_size;
		uint32_t animated_rotation_bit_size;
		uint32_t animated_translation_bit_size;
		uint32_t segment_data;				// offset from the transform_tracks_header
	};

	struct stripped_segment_header : segment_header
	{
		uint32_t sample_indices;			// MSB = first sample of the segment
	};

	struct transform_tracks_header
	{
		uint32_t num_segments;
		uint32_t num_animated_variable_sub_tracks;		// rotations padded to a multiple of 4
		uint32_t num_animated_rotation_sub_tracks;
		uint32_t num_animated_translation_sub_tracks;
		uint32_t num_animated_scale_sub_tracks;
		uint32_t num_constant_rotation_samples;
		uint32_t num_constant_translation_samples;
		uint32_t num_constant_scale_samples;
		uint32_t database_header_offset;				// all offsets relative to this struct
		uint32_t segment_headers_offset;
		uint32_t sub_track_types_offset;
		uint32_t constant_track_data_offset;
		uint32_t clip_range_data_offset;
	};

	// Header of scalar track lists, follows the tracks_header (core/impl/compressed_headers.h:133-165). One bit rate byte per
	// track; constant values; range values (min[C], extent[C]) of the quantized tracks; animated values frame major, MSB first.
	struct scalar_tracks_header
	{
		uint32_t num_bits_per_frame;
		uint32_t metadata_per_track;		// offsets relative to this struct
		uint32_t track_constant_values;
		uint32_t track_range_values;
		uint32_t track_animated_values;
	};

	// bit rate -> bits per component (core/impl/variable_bit_rates.h:42-45); 0 = constant track, 32 = raw fp32
	constexpr uint8_t k_bit_rate_num_bits_v0[] = { 0, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 32 };		// v02_00_00
	constexpr uint8_t k_bit_rate_num_bits[] = { 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22, 23, 32 };

	ACLHIP_HD uint32_t scalar_track_num_components(uint8_t track_type)
	{
		return track_type == k_track_type_float1f ? 1u : track_type == k_track_type_float2f ? 2u : track_type == k_track_type_float3f ? 3u
			: (track_type == k_track_type_float4f || track_type == k_track_type_vector4f) ? 4u : 0u;
	}

	// Optional metadata, the LAST 20 bytes of a blob whose header says has_metadata (core/impl/compressed_headers.h:367-393,
	// compressed_tracks.impl.h:62-65); every offset is from the START of the blob, 0xFFFFFFFF = not stored
	struct optional_metadata_header
	{
		uint32_t track_list_name;
		uint32_t track_name_offsets;
		uint32_t parent_track_indices;		// u32[num_tracks], 0xFFFFFFFF = no parent
		uint32_t track_descriptions;		// transform tracks: 5 / 15 / 12 floats per track for v02_00 / v02_01_99 / later (compressed_tracks.impl.h:237-271)
		uint32_t contributing_error;
	};

	struct tracks_database_header
	{
		uint32_t clip_header_offset;		// into the database's runtime clip/segment header block
	};

	// Runtime (mutable) database metadata, one clip header followed by one segment header per segment
	struct database_runtime_clip_header
	{
		uint32_t clip_hash;
		uint32_t padding;
	};

	struct database_runtime_segment_header
	{
		// (sample offset << 32) | sample indices, [0] = medium importance tier, [1] = low importance tier
		uint64_t tier_metadata[2];
	};

	// ---- compressed_database (core/impl/compressed_headers.h:447-606), follows a raw_buffer_header ----
	struct database_header
	{
		uint32_t tag;							// k_tag_compressed_database
		uint16_t version;
		uint16_t misc_packed;					// bit 0: bulk data inline
		uint32_t num_chunks[2];					// [0] medium importance tier, [1] low importance tier
		uint32_t max_chunk_size;
		uint32_t num_clips;
		uint32_t num_segments;
		uint32_t clip_metadata_offset;			// offsets are relative to this header
		uint32_t bulk_data_size[2];
		uint32_t bulk_data_offset[2];			// k_invalid_offset when the bulk data isn't inline
		uint32_t bulk_data_hash[2];
		// chunk descriptions follow: num_chunks[0] for the medium tier, then num_chunks[1] for the low tier
	};

	struct database_chunk_description
	{
		uint32_t size;
		uint32_t offset;						// of the chunk header, from the start of the tier's bulk data
	};

	struct database_clip_metadata
	{
		uint32_t clip_hash;
		uint32_t clip_header_offset;			// into the runtime clip/segment header block
	};

	struct database_chunk_header
	{
		uint32_t index;
		uint32_t size;
		uint32_t num_segments;
		// database_chunk_segment_header[num_segments] follow, then the sample data
	};

	struct database_chunk_segment_header
	{
		uint32_t clip_hash;
		uint32_t sample_indices;				// samples of the segment stored in this chunk, MSB = first sample
		uint32_t samples_offset;				// from the start of the tier's bulk data
		uint32_t clip_header_offset;			// runtime headers to patch when the chunk streams in / out
		uint32_t segment_header_offset;
	};

	static_assert(sizeof(database_header) == 56, "layout");
	static_assert(sizeof(database_chunk_description) == 8, "layout");
	static_assert(sizeof(database_clip_metadata) == 8, "layout");
	static_assert(sizeof(database_chunk_header) == 12, "layout");
	static_assert(sizeof(database_chunk_segment_header) == 20, "layout");

	static_assert(sizeof(scalar_tracks_header) == 20, "layout");
	static_assert(sizeof(raw_buffer_header) == 8, "layout");
	static_assert(sizeof(tracks_header) == 24, "layout");
	static_assert(sizeof(segment_header) == 16, "layout");
	static_assert(sizeof(stripped_segment_header) == 20, "layout");
	static_assert(sizeof(transform_tracks_header) == 52, "layout");
	static_assert(sizeof(database_runtime_clip_header) == 8, "layout");
	static_assert(sizeof(database_runtime_segment_header) == 16, "layout");

	constexpr uint32_t k_tracks_header_offset = 8;
	constexpr uint32_t k_transform_header_offset = 32;		// = sizeof(raw_buffer_header) + sizeof(tracks_header)
	constexpr uint32_t k_segment_start_indices_offset = 52;	// relative to the transform_tracks_header

	ACLHIP_HD uint32_t align_to_u32(uint32_t value, uint32_t alignment) { return (value + (alignment - 1)) & ~(alignment - 1); }

	// core/hash.h:86-99
	inline uint32_t hash32(const void* data, uint64_t size)
	{
		const uint8_t* bytes = static_cast<const uint8_t*>(data);
		uint32_t acc = 2166136261u;
		for (uint64_t i = 0; i < size; ++i)
			acc = (acc ^ bytes[i]) * 16777619u;
		return acc;
	}

	// core/impl/time_utils.impl.h:102-112 with compressed_tracks::get_finite_duration (core/impl/compressed_tracks.impl.h:102-122)
	inline float finite_duration(const tracks_header& header, uint8_t looping_policy)
	{
		if (looping_policy == k_loop_as_compressed)
			looping_policy = (header.version > k_version_first && header.is_wrap_optimized()) ? k_loop_wrap : k_loop_clamp;

		uint32_t num_samples = header.num_samples;
		if (looping_policy == k_loop_wrap && num_samples != 0)
			num_samples++;

		if (num_samples <= 1)
			return 0.0f;

		return float(num_samples - 1) / header.sample_rate;
	}
}
