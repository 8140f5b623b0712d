// aclhip.hpp -- host side C++ mirror of the reference's decompression surface on top of the C ABI (include/aclhip.h).
//
// The reference's boundary is a header-only template class, acl::decompression_context<settings>
// (/root/reference/includes/acl/decompression/decompress.h:76-201), fed by a user supplied acl::track_writer
// (core/track_writer.h:82-216). This header offers the same names, argument meaning and error behaviour:
//
//     aclhip::device gpu(0);
//     aclhip::decompression_context<aclhip::default_transform_decompression_settings> context;
//     context.initialize(gpu, compressed_tracks_bytes);      // reference: initialize(const compressed_tracks&)
//     context.seek(sample_time, aclhip::sample_rounding_policy::none);
//     context.decompress_tracks(writer);                      // writer.write_rotation/translation/scale, same phase order
//     context.decompress_track(bone_index, writer);
//
// A context decodes a batch of ONE instance per call through aclhip_decompress_tracks_host: it exists so that code written
// against the reference keeps compiling and can be validated; throughput comes from aclhip_decompress_tracks_batch.
// Like the reference: initialize()/relocated() return bool, everything else returns void and silently does nothing on misuse
// (uninitialised context, no seek yet, bad track index: impl/decompress.impl.h:213-214,226-227; impl/decompression.transform.h:1536-1537,1767-1768).
#pragma once

#include <stdint.h>
#include <string.h>

#include <type_traits>
#include <vector>

#include "../../include/aclhip.h"

namespace aclhip
{
	// acl::sample_rounding_policy (core/sample_rounding_policy.h)
	enum class sample_rounding_policy : uint8_t { none = 0, floor = 1, ceil = 2, nearest = 3, per_track = 4 };
	// acl::sample_looping_policy (core/sample_looping_policy.h)
	enum class sample_looping_policy : uint8_t { clamp = 0, wrap = 1, non_looping = clamp, as_compressed = 2 };
	// acl::rotation_normalization_policy_t (decompression/decompression_settings.h:50-62)
	enum class rotation_normalization_policy_t : uint8_t { never = 0, lerp_only = 1, always = 2 };
	// acl::default_sub_track_mode (core/track_writer.h:49-74)
	enum class default_sub_track_mode { skipped, constant, variable, legacy };
	// reference: core/additive_utils.h:43-68
	enum class additive_clip_format8 : uint8_t { none = 0, relative = 1, additive0 = 2, additive1 = 3 };

	// acl::rotation_format8 / vector_format8 (core/track_formats.h:48-71)
	enum class rotation_format8 : uint8_t { quatf_full = 0, quatf_drop_w_full = 2, quatf_drop_w_variable = 3 };
	enum class vector_format8 : uint8_t { vector3f_full = 0, vector3f_variable = 1 };

	struct quatf { float x, y, z, w; };
	struct vector4f { float x, y, z, w; };
	struct qvvf { quatf rotation; vector4f translation; vector4f scale; };		// 48 bytes, the layout of a pose row

	// acl::decompression_settings (decompression/decompression_settings.h:74-166): the switches that matter to the transform path
	struct decompression_settings
	{
		static constexpr bool clamp_sample_time() { return true; }
		static constexpr rotation_normalization_policy_t get_rotation_normalization_policy() { return rotation_normalization_policy_t::always; }
		static constexpr bool skip_initialize_safety_checks() { return false; }
		static constexpr bool is_wrapping_supported() { return true; }
		static constexpr bool is_per_track_rounding_supported() { return true; }
		// which packed formats the settings take (decompression_settings.h:108-112): a context whose settings do not take a clip's formats
		// does not initialize (the reference asserts, decompression.transform.h:102-104)
		static constexpr bool is_rotation_format_supported(rotation_format8 /*format*/) { return true; }
		static constexpr bool is_translation_format_supported(vector_format8 /*format*/) { return true; }
		static constexpr bool is_scale_format_supported(vector_format8 /*format*/) { return true; }
	};

	// acl::debug_transform_decompression_settings (decompression_settings.h:186-191)
	struct debug_transform_decompression_settings : public decompression_settings {};

	// acl::default_transform_decompression_settings (decompression_settings.h:211-232)
	struct default_transform_decompression_settings : public decompression_settings
	{
		static constexpr rotation_normalization_policy_t get_rotation_normalization_policy() { return rotation_normalization_policy_t::lerp_only; }
		static constexpr bool is_per_track_rounding_supported() { return false; }
		static constexpr bool is_rotation_format_supported(rotation_format8 format) { return format == rotation_format8::quatf_drop_w_variable; }
		static constexpr bool is_translation_format_supported(vector_format8 format) { return format == vector_format8::vector3f_variable; }
		static constexpr bool is_scale_format_supported(vector_format8 format) { return format == vector_format8::vector3f_variable; }
	};

	// The optional metadata of a blob, as compressed_tracks hands it out (core/impl/compressed_tracks.impl.h:175-275). Host only.
	static constexpr uint32_t k_invalid_track_index = 0xFFFFFFFFu;
	struct track_desc_transformf			// core/track_desc.h:86-158, what a compressed blob keeps of it
	{
		qvvf default_value = qvvf{ quatf{ 0.0f, 0.0f, 0.0f, 1.0f }, vector4f{ 0.0f, 0.0f, 0.0f, 0.0f }, vector4f{ 1.0f, 1.0f, 1.0f, 0.0f } };
		uint32_t output_index = k_invalid_track_index;
		uint32_t parent_index = k_invalid_track_index;
		float precision = 0.01f;
		float shell_distance = 3.0f;
	};

	// compressed_tracks::get_parent_track_index (compressed_tracks.impl.h:175-190): k_invalid_track_index for a root, and when the blob
	// does not store parent indices
	inline uint32_t get_parent_track_index(const void* compressed_tracks, uint64_t size, uint32_t track_index)
	{
		aclhip_clip_metadata_info info = {};
		uint32_t num_tracks = 0;
		if (compressed_tracks == nullptr || size < 32)
			return k_invalid_track_index;
		memcpy(&num_tracks, static_cast<const uint8_t*>(compressed_tracks) + 16, 4);
		if (track_index >= num_tracks)
			return k_invalid_track_index;
		std::vector<uint32_t> parents(num_tracks, k_invalid_track_index);
		if (aclhip_read_clip_metadata(compressed_tracks, size, &info, parents.data(), nullptr, nullptr, nullptr, num_tracks) != ACLHIP_OK || info.has_parent_track_indices == 0)
			return k_invalid_track_index;
		return parents[track_index];
	}

	// compressed_tracks::get_track_description(track_index, track_desc_transformf&) (compressed_tracks.impl.h:214-275)
	inline bool get_track_description(const void* compressed_tracks, uint64_t size, uint32_t track_index, track_desc_transformf& out_description)
	{
		aclhip_clip_metadata_info info = {};
		uint32_t num_tracks = 0;
		if (compressed_tracks == nullptr || size < 32)
			return false;
		memcpy(&num_tracks, static_cast<const uint8_t*>(compressed_tracks) + 16, 4);
		if (track_index >= num_tracks)
			return false;
		std::vector<uint32_t> parents(num_tracks, k_invalid_track_index);
		std::vector<float> defaults(size_t(num_tracks) * 12), precisions(num_tracks), shells(num_tracks);
		if (aclhip_read_clip_metadata(compressed_tracks, size, &info, parents.data(), defaults.data(), precisions.data(), shells.data(), num_tracks) != ACLHIP_OK
			|| info.has_track_descriptions == 0)
			return false;
		memcpy(&out_description.default_value, &defaults[size_t(track_index) * 12], sizeof(qvvf));
		out_description.output_index = track_index;
		out_description.parent_index = parents[track_index];
		out_description.precision = precisions[track_index];
		out_description.shell_distance = shells[track_index];
		return true;
	}

	// acl::track_writer (core/track_writer.h:82-216), transform part. Derive and override what you need.
	struct track_writer
	{
		sample_rounding_policy get_rounding_policy(sample_rounding_policy seek_policy, uint32_t /*track_index*/) const { return seek_policy; }

		static constexpr default_sub_track_mode get_default_rotation_mode() { return default_sub_track_mode::constant; }
		static constexpr default_sub_track_mode get_default_translation_mode() { return default_sub_track_mode::constant; }
		static constexpr default_sub_track_mode get_default_scale_mode() { return default_sub_track_mode::legacy; }

		quatf get_constant_default_rotation() const { return quatf{ 0.0f, 0.0f, 0.0f, 1.0f }; }
		vector4f get_constant_default_translation() const { return vector4f{ 0.0f, 0.0f, 0.0f, 0.0f }; }
		vector4f get_constant_default_scale() const { return vector4f{ 1.0f, 1.0f, 1.0f, 0.0f }; }

		quatf get_variable_default_rotation(uint32_t /*track_index*/) const { return quatf{ 0.0f, 0.0f, 0.0f, 1.0f }; }
		vector4f get_variable_default_translation(uint32_t /*track_index*/) const { return vector4f{ 0.0f, 0.0f, 0.0f, 0.0f }; }
		vector4f get_variable_default_scale(uint32_t /*track_index*/) const { return vector4f{ 1.0f, 1.0f, 1.0f, 0.0f }; }

		static constexpr bool skip_all_rotations() { return false; }
		static constexpr bool skip_all_translations() { return false; }
		static constexpr bool skip_all_scales() { return false; }

		// scalar track lists (core/track_writer.h:101-158): value.x .. value.w hold the track's 1 / 2 / 3 / 4 components
		bool skip_track_float1(uint32_t /*track_index*/) const { return false; }
		bool skip_track_float2(uint32_t /*track_index*/) const { return false; }
		bool skip_track_float3(uint32_t /*track_index*/) const { return false; }
		bool skip_track_float4(uint32_t /*track_index*/) const { return false; }
		bool skip_track_vector4(uint32_t /*track_index*/) const { return false; }
		void write_float1(uint32_t /*track_index*/, float /*value*/) {}
		void write_float2(uint32_t /*track_index*/, vector4f /*value*/) {}
		void write_float3(uint32_t /*track_index*/, vector4f /*value*/) {}
		void write_float4(uint32_t /*track_index*/, vector4f /*value*/) {}
		void write_vector4(uint32_t /*track_index*/, vector4f /*value*/) {}

		bool skip_track_rotation(uint32_t /*track_index*/) const { return false; }
		bool skip_track_translation(uint32_t /*track_index*/) const { return false; }
		bool skip_track_scale(uint32_t /*track_index*/) const { return false; }

		void write_rotation(uint32_t /*track_index*/, quatf /*rotation*/) {}
		void write_translation(uint32_t /*track_index*/, vector4f /*translation*/) {}
		void write_scale(uint32_t /*track_index*/, vector4f /*scale*/) {}
	};

	// RAII owner of an aclhip_context (one per GPU); clips registered by contexts live in it.
	class device
	{
	public:
		// (a library built from another include/aclhip.h would read this header's structs wrongly: no context then)
		explicit device(int device_index = 0) { if (aclhip_abi_version() != ACLHIP_ABI_VERSION || aclhip_create(device_index, &m_context) != ACLHIP_OK) m_context = nullptr; }
		~device() { aclhip_destroy(m_context); }
		device(const device&) = delete;
		device& operator=(const device&) = delete;

		bool is_valid() const { return m_context != nullptr; }
		aclhip_context* get() const { return m_context; }

	private:
		aclhip_context* m_context = nullptr;
	};

	namespace impl
	{
		inline aclhip_default_mode to_c_mode(default_sub_track_mode mode)
		{
			switch (mode)
			{
			case default_sub_track_mode::skipped: return ACLHIP_DEFAULT_SKIPPED;
			case default_sub_track_mode::constant: return ACLHIP_DEFAULT_CONSTANT;
			case default_sub_track_mode::variable: return ACLHIP_DEFAULT_VARIABLE;
			default: return ACLHIP_DEFAULT_LEGACY;
			}
		}

		// Sub-track class of a track (core/impl/compressed_headers.h:214-224): 0 default, 1 constant, 2 animated
		inline uint32_t sub_track_class(const uint8_t* blob, uint32_t kind, uint32_t track_index)
		{
			uint32_t num_tracks, misc_packed, types_offset;
			memcpy(&num_tracks, blob + 16, 4);
			memcpy(&misc_packed, blob + 28, 4);
			memcpy(&types_offset, blob + 32 + 40, 4);			// transform_tracks_header::sub_track_types_offset
			if (kind == 2 && (misc_packed & 1u) == 0)
				return 0;											// no scale: every scale sub-track is default
			const uint32_t num_entries = (num_tracks + 15) / 16;
			uint32_t packed;
			memcpy(&packed, blob + 32 + types_offset + 4 * (size_t(kind) * num_entries + track_index / 16), 4);
			return (packed >> ((15 - (track_index % 16)) * 2)) & 3u;
		}
	}

	// decompression_settings.h:172-204: scalar track lists; the context dispatches on the bound clip's track type either way
	struct debug_scalar_decompression_settings : public decompression_settings {};
	struct default_scalar_decompression_settings : public decompression_settings
	{
		static constexpr bool is_per_track_rounding_supported() { return false; }
	};

	// core/quality_tiers.h: the highest importance tier lives in the compressed_tracks, the other two in a compressed_database
	enum class quality_tier : uint8_t { highest_importance = 0, medium_importance = 1, lowest_importance = 2 };

	// decompression/database/database.h:48-73
	enum class database_stream_request_result { done, dispatched, streaming_in_progress, context_not_initialized, invalid_database_tier, no_free_streaming_requests };

	struct database_settings {};
	struct default_database_settings : public database_settings {};

	// Mirrors acl::database_context<settings> (decompression/database/database.h:69-201) over a database registered with the GPU.
	// The reference takes an allocator and two database_streamer objects; here the streamer is built in: bulk data is copied once
	// to pinned host memory at initialize() and stream_in() is a hipMemcpyAsync into HBM followed by the metadata update, both on
	// `stream`. Requests complete in stream order, so is_streaming() is always false and is_streamed_in() reflects what has been
	// enqueued.
	template<class database_settings_type>
	class database_context
	{
	public:
		using settings_type = database_settings_type;

		database_context() = default;
		~database_context() { reset(); }
		database_context(const database_context&) = delete;
		database_context& operator=(const database_context&) = delete;

		// reference: initialize(allocator, database) for inline bulk data (database.h:110); `size` = database.get_size()
		bool initialize(device& gpu, const void* compressed_database, uint64_t size) { return initialize(gpu, compressed_database, size, nullptr, nullptr); }

		// reference: initialize(allocator, database, medium_tier_streamer, low_tier_streamer) (database.h:116): the streamers' bulk data
		bool initialize(device& gpu, const void* compressed_database, uint64_t size, const void* bulk_data_medium, const void* bulk_data_low)
		{
			reset();
			if (!gpu.is_valid() || compressed_database == nullptr)
				return false;
			aclhip_database handle = ACLHIP_INVALID_HANDLE;
			// is_valid(false): no hash check, like database.impl.h:113
			if (aclhip_register_database(gpu.get(), compressed_database, size, bulk_data_medium, bulk_data_low, 0, &handle) != ACLHIP_OK)
				return false;
			m_device = &gpu;
			m_database = handle;
			m_compressed_database = compressed_database;
			return true;
		}

		// A database whose bulk data is served by the caller's own streamers (aclhip_register_database_streamed): stream_in_from()
		// hands the tier's bulk data over with every request
		bool initialize_streamed(device& gpu, const void* compressed_database, uint64_t size)
		{
			reset();
			if (!gpu.is_valid() || compressed_database == nullptr)
				return false;
			aclhip_database handle = ACLHIP_INVALID_HANDLE;
			if (aclhip_register_database_streamed(gpu.get(), compressed_database, size, 0, &handle) != ACLHIP_OK)
				return false;
			m_device = &gpu;
			m_database = handle;
			m_compressed_database = compressed_database;
			return true;
		}

		database_stream_request_result stream_in_from(quality_tier tier, uint32_t num_chunks_to_stream, const void* tier_bulk_data, void* stream = nullptr)
		{
			if (!is_initialized())
				return database_stream_request_result::context_not_initialized;
			if (tier == quality_tier::highest_importance)
				return database_stream_request_result::invalid_database_tier;
			uint32_t moved = 0;
			if (aclhip_database_stream_in_from(m_device->get(), m_database, uint32_t(tier), num_chunks_to_stream, tier_bulk_data, stream, &moved) != ACLHIP_OK)
				return database_stream_request_result::no_free_streaming_requests;
			return moved != 0 ? database_stream_request_result::dispatched : database_stream_request_result::done;
		}

		const void* get_compressed_database() const { return m_compressed_database; }
		bool is_initialized() const { return m_device != nullptr; }

		// Fails (and stays bound) while decompression contexts are bound to it; the reference leaves that to the caller
		void reset()
		{
			if (m_device != nullptr && aclhip_unregister_database(m_device->get(), m_database) != ACLHIP_OK)
				return;
			m_device = nullptr;
			m_database = ACLHIP_INVALID_HANDLE;
			m_compressed_database = nullptr;
		}

		bool is_bound_to(const void* compressed_database) const { return is_initialized() && compressed_database == m_compressed_database; }

		// reference: contains(const compressed_tracks&) (database.h:147): the clip's hash is listed by the database
		bool contains(const void* compressed_tracks) const
		{
			if (!is_initialized() || compressed_tracks == nullptr)
				return false;
			const uint8_t* tracks = static_cast<const uint8_t*>(compressed_tracks);
			const uint8_t* db = static_cast<const uint8_t*>(m_compressed_database);
			uint32_t misc_packed, clip_hash, num_clips, clip_metadata_offset;
			memcpy(&misc_packed, tracks + 28, 4);
			if ((misc_packed & (1u << 8)) == 0)
				return false;		// not bound to any database
			memcpy(&clip_hash, tracks + 4, 4);
			memcpy(&num_clips, db + 8 + 20, 4);
			memcpy(&clip_metadata_offset, db + 8 + 28, 4);
			for (uint32_t i = 0; i < num_clips; ++i)
			{
				uint32_t hash;
				memcpy(&hash, db + 8 + clip_metadata_offset + size_t(i) * 8, 4);
				if (hash == clip_hash)
					return true;
			}
			return false;
		}

		bool is_streamed_in(quality_tier tier) const
		{
			aclhip_database_info info;
			if (!is_initialized() || tier == quality_tier::highest_importance || aclhip_get_database_info(m_device->get(), m_database, &info) != ACLHIP_OK)
				return is_initialized() && tier == quality_tier::highest_importance;
			const uint32_t tier_index = uint32_t(tier) - 1;
			return info.num_loaded_chunks[tier_index] == info.num_chunks[tier_index];
		}

		bool is_streaming(quality_tier /*tier*/) const { return false; }

		database_stream_request_result stream_in(quality_tier tier, uint32_t num_chunks_to_stream = ~0u, void* stream = nullptr) { return request(tier, num_chunks_to_stream, stream, true); }
		database_stream_request_result stream_out(quality_tier tier, uint32_t num_chunks_to_stream = ~0u, void* stream = nullptr) { return request(tier, num_chunks_to_stream, stream, false); }

		device* get_device() const { return m_device; }
		aclhip_database get_handle() const { return m_database; }

	private:
		static_assert(std::is_base_of<database_settings, settings_type>::value, "database_settings_type must derive from database_settings!");

		database_stream_request_result request(quality_tier tier, uint32_t num_chunks, void* stream, bool in)
		{
			if (!is_initialized())
				return database_stream_request_result::context_not_initialized;
			if (tier == quality_tier::highest_importance)
				return database_stream_request_result::invalid_database_tier;
			uint32_t moved = 0;
			const aclhip_status status = in ? aclhip_database_stream_in(m_device->get(), m_database, uint32_t(tier), num_chunks, stream, &moved)
				: aclhip_database_stream_out(m_device->get(), m_database, uint32_t(tier), num_chunks, stream, &moved);
			if (status != ACLHIP_OK)
				return database_stream_request_result::no_free_streaming_requests;
			return moved != 0 ? database_stream_request_result::dispatched : database_stream_request_result::done;
		}

		device* m_device = nullptr;
		aclhip_database m_database = ACLHIP_INVALID_HANDLE;
		const void* m_compressed_database = nullptr;
	};

	template<class decompression_settings_type>
	class decompression_context
	{
	public:
		using settings_type = decompression_settings_type;

		decompression_context() = default;
		~decompression_context() { reset(); }
		decompression_context(const decompression_context&) = delete;
		decompression_context& operator=(const decompression_context&) = delete;

		// reference: bool initialize(const compressed_tracks&) (decompress.h:103). The bytes are copied to the GPU; `compressed_tracks`
		// must stay alive while bound because default / constant / animated classes are read from it when replaying writes.
		bool initialize(device& gpu, const void* compressed_tracks, uint64_t size)
		{
			reset();
			if (!gpu.is_valid() || compressed_tracks == nullptr)
				return false;

			if (!formats_are_supported(compressed_tracks, size))
				return false;
			aclhip_clip clip = ACLHIP_INVALID_HANDLE;
			// is_valid(false): the reference does not check the hash on initialize (impl/decompress.impl.h:70)
			if (aclhip_register_clip(gpu.get(), compressed_tracks, size, 0, &clip) != ACLHIP_OK)
				return false;

			m_device = &gpu;
			m_clip = clip;
			m_tracks = static_cast<const uint8_t*>(compressed_tracks);
			aclhip_get_clip_info(gpu.get(), clip, &m_info);
			m_looping_policy = settings_type::is_wrapping_supported() ? static_cast<sample_looping_policy>(m_info.looping_policy) : sample_looping_policy::clamp;
			m_sample_time = -1.0f;
			return true;
		}

		// reference: bool initialize(const compressed_tracks&, const database_context<..>&) (decompress.h:108, impl/decompress.impl.h:85-113):
		// fails when the database context is not initialized or does not contain the clip
		template<class database_settings_type>
		bool initialize(const void* compressed_tracks, uint64_t size, const database_context<database_settings_type>& database)
		{
			reset();
			if (compressed_tracks == nullptr || !database.is_initialized() || !database.contains(compressed_tracks) || !formats_are_supported(compressed_tracks, size))
				return false;

			device& gpu = *database.get_device();
			aclhip_clip clip = ACLHIP_INVALID_HANDLE;
			if (aclhip_register_clip_with_database(gpu.get(), compressed_tracks, size, 0, database.get_handle(), &clip) != ACLHIP_OK)
				return false;

			m_device = &gpu;
			m_clip = clip;
			m_tracks = static_cast<const uint8_t*>(compressed_tracks);
			aclhip_get_clip_info(gpu.get(), clip, &m_info);
			m_looping_policy = settings_type::is_wrapping_supported() ? static_cast<sample_looping_policy>(m_info.looping_policy) : sample_looping_policy::clamp;
			m_sample_time = -1.0f;
			return true;
		}

		bool is_initialized() const { return m_device != nullptr; }

		// the settings' is_*_format_supported against the blob's packed formats (transform clips; decompression.transform.h:95-104)
		static bool formats_are_supported(const void* compressed_tracks, uint64_t size)
		{
			if (size < 32)
				return true;		// (registration refuses it with its own message)
			const uint8_t* blob = static_cast<const uint8_t*>(compressed_tracks);
			uint32_t num_tracks, misc_packed;
			memcpy(&num_tracks, blob + 16, 4);
			memcpy(&misc_packed, blob + 28, 4);
			if (blob[15] != 12 || num_tracks == 0)
				return true;		// scalar track lists have no packed formats
			const bool has_scale = (misc_packed & 1u) != 0;
			return settings_type::is_rotation_format_supported(static_cast<rotation_format8>((misc_packed >> 4) & 15u))
				&& settings_type::is_translation_format_supported(static_cast<vector_format8>((misc_packed >> 3) & 1u))
				&& (!has_scale || settings_type::is_scale_format_supported(static_cast<vector_format8>((misc_packed >> 2) & 1u)));
		}

		void reset()
		{
			if (m_device != nullptr)
				aclhip_unregister_clip(m_device->get(), m_clip);
			m_device = nullptr;
			m_tracks = nullptr;
			m_clip = ACLHIP_INVALID_HANDLE;
		}

		// reference: relocated(const compressed_tracks&) (decompress.h:123): same clip, new address
		bool relocated(const void* compressed_tracks)
		{
			if (!is_bound_to_hash(compressed_tracks))
				return false;
			m_tracks = static_cast<const uint8_t*>(compressed_tracks);
			m_sample_time = -1.0f;		// forces a new seek, like relocated_v0 (impl/decompression.transform.h:151-154)
			return true;
		}

		// reference: is_bound_to(const compressed_tracks&) (decompress.h:138): same pointer and same hash
		bool is_bound_to(const void* compressed_tracks) const { return compressed_tracks == m_tracks && is_bound_to_hash(compressed_tracks); }

		void set_looping_policy(sample_looping_policy policy)
		{
			if (!is_initialized() || !settings_type::is_wrapping_supported())
				return;		// only clamping is supported (impl/decompression.transform.h:189-190)
			if (policy == sample_looping_policy::as_compressed)
				policy = static_cast<sample_looping_policy>(m_info.looping_policy);
			m_looping_policy = policy;
		}

		sample_looping_policy get_looping_policy() const { return m_looping_policy; }

		// reference: seek(float sample_time, sample_rounding_policy) (decompress.h:160). The seek itself runs on the GPU with the decode.
		void seek(float sample_time, sample_rounding_policy rounding_policy)
		{
			if (!is_initialized())
				return;
			if (rounding_policy == sample_rounding_policy::per_track && !settings_type::is_per_track_rounding_supported())
				return;		// the reference asserts here
			m_sample_time = sample_time < 0.0f && settings_type::clamp_sample_time() ? 0.0f : sample_time;
			m_rounding_policy = rounding_policy;
		}

		// ---- pose consumers: what reference callers do next with the pose decompress_tracks wrote, fused into the decode ----
		// Parent of every transform (track_desc_transformf::parent_index), sorted parent first; k_no_parent marks roots.
		static constexpr uint32_t k_no_parent = ACLHIP_NO_PARENT;
		bool set_parent_indices(const uint32_t* parent_indices, uint32_t num_transforms)
		{
			return is_initialized() && aclhip_set_clip_hierarchy(m_device->get(), m_clip, parent_indices, num_transforms) == ACLHIP_OK;
		}
		// ... the parent indices the blob itself carries (compressed_tracks::get_parent_track_index); false when it carries none
		bool set_parent_indices_from_metadata()
		{
			return is_initialized() && aclhip_set_clip_hierarchy_from_metadata(m_device->get(), m_clip) == ACLHIP_OK;
		}

		// The pose at the last seek() as a caller of the reference would have it after
		//     decompress_tracks(writer into local_pose);
		//     local_pose[i] = acl::apply_additive_to_base(additive_format, base_pose[i], local_pose[i]);   (core/additive_utils.h:150)
		//     acl::local_to_object_space(parent_indices, local_pose, num_transforms, out_pose);            (compression/transform_pose_utils.h:35)
		// with the track_writer's default sub-track modes. `base` is another context of the same device, sought to the base's sample
		// time (null with additive_clip_format8::none). out_pose: one qvvf per track of the clip. False when nothing was written.
		bool decompress_pose(qvvf* out_pose, bool object_space, additive_clip_format8 additive_format = additive_clip_format8::none,
			const decompression_context* base = nullptr)
		{
			if (!is_initialized() || m_info.track_type != 12 || m_info.num_tracks == 0 || m_sample_time < 0.0f || out_pose == nullptr)
				return false;
			if (additive_format != additive_clip_format8::none && (base == nullptr || !base->is_initialized() || base->m_device != m_device || base->m_sample_time < 0.0f))
				return false;

			aclhip_decompress_params params;
			aclhip_default_params(&params);
			params.rounding_policy = static_cast<uint8_t>(m_rounding_policy);
			params.looping_policy = static_cast<uint8_t>(m_looping_policy);
			params.normalization = static_cast<uint8_t>(settings_type::get_rotation_normalization_policy());

			aclhip_pose_consumers consumers = {};
			consumers.additive_format = static_cast<uint32_t>(additive_format);
			consumers.object_space = object_space ? 1 : 0;
			float base_time = 0.0f;
			if (additive_format != additive_clip_format8::none)
			{
				base_time = base->m_sample_time;
				consumers.base_clips = &base->m_clip;
				consumers.base_sample_times = &base_time;
			}
			const float sample_time = m_sample_time;
			const uint64_t before = rejected_count();
			if (aclhip_decompress_poses_host(m_device->get(), &m_clip, &sample_time, 1, &params, &consumers, out_pose, uint64_t(m_info.num_tracks) * sizeof(qvvf)) != ACLHIP_OK)
				return false;
			return rejected_count() == before;
		}

		// The weighted blend of this context's pose (at its last seek()) with the poses of `num_others` further contexts of the same
		// device (each at ITS last seek(); same number of tracks), then local -> object space on request: what an engine's locomotion
		// blend does with K decompress_tracks results. weights[0] belongs to this context, weights[1 + k] to others[k]; used as given.
		// The reference ships no blend function; the operation order is stated next to aclhip_pose_consumers::num_blend_clips
		// (include/aclhip.h). False when nothing was written.
		bool decompress_blended_pose(qvvf* out_pose, const decompression_context* const* others, uint32_t num_others, const float* weights, bool object_space)
		{
			if (!is_initialized() || m_info.track_type != 12 || m_info.num_tracks == 0 || m_sample_time < 0.0f || out_pose == nullptr
				|| others == nullptr || weights == nullptr || num_others == 0 || num_others + 1 > ACLHIP_MAX_BLEND_CLIPS)
				return false;
			aclhip_clip other_clips[ACLHIP_MAX_BLEND_CLIPS];
			float other_times[ACLHIP_MAX_BLEND_CLIPS];
			for (uint32_t k = 0; k < num_others; ++k)
			{
				if (others[k] == nullptr || !others[k]->is_initialized() || others[k]->m_device != m_device || others[k]->m_sample_time < 0.0f)
					return false;
				other_clips[k] = others[k]->m_clip;
				other_times[k] = others[k]->m_sample_time;
			}

			aclhip_decompress_params params;
			aclhip_default_params(&params);
			params.rounding_policy = static_cast<uint8_t>(m_rounding_policy);
			params.looping_policy = static_cast<uint8_t>(m_looping_policy);
			params.normalization = static_cast<uint8_t>(settings_type::get_rotation_normalization_policy());

			aclhip_pose_consumers consumers = {};
			consumers.object_space = object_space ? 1 : 0;
			consumers.num_blend_clips = num_others + 1;
			consumers.blend_clips = other_clips;
			consumers.blend_sample_times = other_times;
			consumers.blend_weights = weights;
			const float sample_time = m_sample_time;
			const uint64_t before = rejected_count();
			if (aclhip_decompress_poses_host(m_device->get(), &m_clip, &sample_time, 1, &params, &consumers, out_pose, uint64_t(m_info.num_tracks) * sizeof(qvvf)) != ACLHIP_OK)
				return false;
			return rejected_count() == before;
		}

		// reference: decompress_tracks(track_writer_type&) (decompress.h:166)
		template<class track_writer_type>
		void decompress_tracks(track_writer_type& writer)
		{
			if (!is_initialized() || m_info.num_tracks == 0 || m_sample_time < 0.0f)
				return;

			if (m_info.track_type != 12)
			{
				// scalar track lists: decompress_tracks_v0 writes the tracks in order (impl/decompression.scalar.h:266-475)
				const uint32_t num_components = m_info.num_components;
				m_pose.assign(size_t(m_info.num_tracks) * num_components, 0.0f);
				if (!run_scalar(writer, nullptr, m_pose.data(), uint64_t(m_info.num_tracks) * num_components * 4))
					return;
				for (uint32_t track = 0; track < m_info.num_tracks; ++track)
					write_scalar_track(writer, track, &m_pose[size_t(track) * num_components], true);
				return;
			}

			const uint32_t num_tracks = m_info.num_tracks;
			m_pose.assign(size_t(num_tracks) * 12, 0.0f);
			if (!run(writer, nullptr, m_pose.data(), num_tracks))
				return;

			// Replay in the reference's phase order (impl/decompression.transform.h:1605-1733): default rotations, constant rotations,
			// default translations, constant translations, default / constant scales, animated rotations, translations, scales
			static const uint32_t phases[8][2] = { { 0, 0 }, { 0, 1 }, { 1, 0 }, { 1, 1 }, { 2, 0 }, { 2, 1 }, { 0, 2 }, { 1, 2 } };
			for (uint32_t phase = 0; phase < 9; ++phase)
			{
				const uint32_t kind = phase < 8 ? phases[phase][0] : 2;
				const uint32_t cls = phase < 8 ? phases[phase][1] : 2;
				for (uint32_t track = 0; track < num_tracks; ++track)
					if (impl::sub_track_class(m_tracks, kind, track) == cls)
						write_sub_track(writer, kind, cls, track, &m_pose[size_t(track) * 12], true);
			}
		}

		// reference: decompress_track(uint32_t, track_writer_type&) (decompress.h:172). Like the reference it does not consult
		// skip_track_* (impl/decompression.transform.h:1985-2046).
		template<class track_writer_type>
		void decompress_track(uint32_t track_index, track_writer_type& writer)
		{
			if (!is_initialized() || m_sample_time < 0.0f || track_index >= m_info.num_tracks)
				return;

			if (m_info.track_type != 12)
			{
				// decompress_track_v0 (impl/decompression.scalar.h:482-715): no skip_track_* consultation
				float value[4] = { 0.0f, 0.0f, 0.0f, 0.0f };
				if (run_scalar(writer, &track_index, value, 16))
					write_scalar_track(writer, track_index, value, false);
				return;
			}

			float transform[12] = { 0 };
			if (!run(writer, &track_index, transform, 1))
				return;

			for (uint32_t kind = 0; kind < 3; ++kind)
				write_sub_track(writer, kind, impl::sub_track_class(m_tracks, kind, track_index), track_index, transform, false);
		}

	private:
		uint64_t rejected_count() const
		{
			uint64_t count = 0;
			(void)aclhip_get_rejected_instance_count(m_device->get(), &count);
			return count;
		}

		bool is_bound_to_hash(const void* compressed_tracks) const
		{
			if (!is_initialized() || compressed_tracks == nullptr)
				return false;
			int matches = 0;
			return aclhip_clip_matches(m_device->get(), m_clip, compressed_tracks, &matches) == ACLHIP_OK && matches != 0;
		}

		template<class track_writer_type>
		bool run_scalar(track_writer_type& writer, const uint32_t* track_index, float* out, uint64_t stride_bytes)
		{
			aclhip_decompress_params params;
			aclhip_default_params(&params);
			params.rounding_policy = static_cast<uint8_t>(m_rounding_policy);
			params.looping_policy = static_cast<uint8_t>(m_looping_policy);
			params.per_track_rounding = settings_type::is_per_track_rounding_supported() ? 1 : 0;
			if (settings_type::is_per_track_rounding_supported())
			{
				m_track_rounding.resize(m_info.num_tracks);
				for (uint32_t i = 0; i < m_info.num_tracks; ++i)
					m_track_rounding[i] = static_cast<uint8_t>(writer.get_rounding_policy(m_rounding_policy, i));
				params.track_rounding_policies = m_track_rounding.data();
			}
			const aclhip_status status = track_index != nullptr
				? aclhip_decompress_scalar_track_host(m_device->get(), &m_clip, &m_sample_time, track_index, 1, &params, out, stride_bytes)
				: aclhip_decompress_scalar_tracks_host(m_device->get(), &m_clip, &m_sample_time, 1, &params, out, stride_bytes);
			return status == ACLHIP_OK;
		}

		template<class track_writer_type>
		void write_scalar_track(track_writer_type& writer, uint32_t track_index, const float* value, bool honour_skips)
		{
			const vector4f packed = { value[0], m_info.num_components > 1 ? value[1] : 0.0f, m_info.num_components > 2 ? value[2] : 0.0f, m_info.num_components > 3 ? value[3] : 0.0f };
			switch (m_info.track_type)
			{
			case 0: if (!honour_skips || !writer.skip_track_float1(track_index)) writer.write_float1(track_index, value[0]); break;
			case 1: if (!honour_skips || !writer.skip_track_float2(track_index)) writer.write_float2(track_index, packed); break;
			case 2: if (!honour_skips || !writer.skip_track_float3(track_index)) writer.write_float3(track_index, packed); break;
			case 3: if (!honour_skips || !writer.skip_track_float4(track_index)) writer.write_float4(track_index, packed); break;
			case 4: if (!honour_skips || !writer.skip_track_vector4(track_index)) writer.write_vector4(track_index, packed); break;
			default: break;
			}
		}

		template<class track_writer_type>
		bool run(track_writer_type& writer, const uint32_t* track_index, float* out, uint32_t out_tracks)
		{
			aclhip_decompress_params params;
			aclhip_default_params(&params);
			params.rounding_policy = static_cast<uint8_t>(m_rounding_policy);
			params.looping_policy = static_cast<uint8_t>(m_looping_policy);
			params.normalization = static_cast<uint8_t>(settings_type::get_rotation_normalization_policy());
			params.per_track_rounding = settings_type::is_per_track_rounding_supported() ? 1 : 0;
			params.default_rotation_mode = impl::to_c_mode(track_writer_type::get_default_rotation_mode());
			params.default_translation_mode = impl::to_c_mode(track_writer_type::get_default_translation_mode());
			params.default_scale_mode = impl::to_c_mode(track_writer_type::get_default_scale_mode());

			const uint32_t num_tracks = m_info.num_tracks;
			const bool any_variable = track_writer_type::get_default_rotation_mode() == default_sub_track_mode::variable
				|| track_writer_type::get_default_translation_mode() == default_sub_track_mode::variable
				|| track_writer_type::get_default_scale_mode() == default_sub_track_mode::variable;

			// default values the writer supplies; a variable mode needs the per track table, a constant one row
			uint32_t default_count = any_variable ? num_tracks : 1;
			m_defaults.assign(size_t(default_count) * 12, 0.0f);
			for (uint32_t i = 0; i < default_count; ++i)
			{
				float* row = &m_defaults[size_t(i) * 12];
				const quatf rotation = track_writer_type::get_default_rotation_mode() == default_sub_track_mode::variable ? writer.get_variable_default_rotation(i) : writer.get_constant_default_rotation();
				const vector4f translation = track_writer_type::get_default_translation_mode() == default_sub_track_mode::variable ? writer.get_variable_default_translation(i) : writer.get_constant_default_translation();
				vector4f scale = track_writer_type::get_default_scale_mode() == default_sub_track_mode::variable ? writer.get_variable_default_scale(i) : writer.get_constant_default_scale();
				if (track_writer_type::get_default_scale_mode() == default_sub_track_mode::legacy)
				{
					const float legacy = float((m_tracks[28] >> 1) & 1u);		// tracks_header default scale bit
					scale = vector4f{ legacy, legacy, legacy, 0.0f };
				}
				row[0] = rotation.x; row[1] = rotation.y; row[2] = rotation.z; row[3] = rotation.w;
				row[4] = translation.x; row[5] = translation.y; row[6] = translation.z;
				row[8] = scale.x; row[9] = scale.y; row[10] = scale.z;
			}

			// with one mode variable every CONSTANT-mode kind must repeat its value on each row; the loop above already did.
			// a legacy scale mode next to user defaults is expressed as constant (the value was resolved above)
			if (params.default_scale_mode == ACLHIP_DEFAULT_LEGACY)
				params.default_scale_mode = any_variable ? ACLHIP_DEFAULT_VARIABLE : ACLHIP_DEFAULT_CONSTANT;
			if (any_variable)
			{
				if (params.default_rotation_mode == ACLHIP_DEFAULT_CONSTANT) params.default_rotation_mode = ACLHIP_DEFAULT_VARIABLE;
				if (params.default_translation_mode == ACLHIP_DEFAULT_CONSTANT) params.default_translation_mode = ACLHIP_DEFAULT_VARIABLE;
				if (params.default_scale_mode == ACLHIP_DEFAULT_CONSTANT) params.default_scale_mode = ACLHIP_DEFAULT_VARIABLE;
			}
			params.default_values = m_defaults.data();

			if (settings_type::is_per_track_rounding_supported())
			{
				m_track_rounding.resize(num_tracks);
				for (uint32_t i = 0; i < num_tracks; ++i)
					m_track_rounding[i] = static_cast<uint8_t>(writer.get_rounding_policy(m_rounding_policy, i));
				params.track_rounding_policies = m_track_rounding.data();
			}

			const float sample_time = m_sample_time;
			if (track_index != nullptr)
				return aclhip_decompress_track_host(m_device->get(), &m_clip, &sample_time, track_index, 1, &params, default_count, out) == ACLHIP_OK;
			// the writer's static output switches (track_writer::skip_all_*, core/track_writer.h:181-183): what it skips is not stored
			aclhip_output_desc output = {};
			output.layout = ACLHIP_LAYOUT_QVV48;
			output.skip_rotations = track_writer_type::skip_all_rotations() ? 1 : 0;
			output.skip_translations = track_writer_type::skip_all_translations() ? 1 : 0;
			output.skip_scales = track_writer_type::skip_all_scales() ? 1 : 0;
			return aclhip_decompress_tracks_host_out(m_device->get(), &m_clip, &sample_time, 1, &params, default_count, &output, out, uint64_t(out_tracks) * 48) == ACLHIP_OK;
		}

		template<class track_writer_type>
		static void write_sub_track(track_writer_type& writer, uint32_t kind, uint32_t cls, uint32_t track_index, const float* qvv, bool honour_skips)
		{
			const default_sub_track_mode mode = kind == 0 ? track_writer_type::get_default_rotation_mode()
				: (kind == 1 ? track_writer_type::get_default_translation_mode() : track_writer_type::get_default_scale_mode());
			if (cls == 0 && mode == default_sub_track_mode::skipped)
				return;		// nothing to write

			if (kind == 0)
			{
				if (honour_skips && (track_writer_type::skip_all_rotations() || writer.skip_track_rotation(track_index)))
					return;
				writer.write_rotation(track_index, quatf{ qvv[0], qvv[1], qvv[2], qvv[3] });
			}
			else if (kind == 1)
			{
				if (honour_skips && (track_writer_type::skip_all_translations() || writer.skip_track_translation(track_index)))
					return;
				writer.write_translation(track_index, vector4f{ qvv[4], qvv[5], qvv[6], 0.0f });
			}
			else
			{
				if (honour_skips && (track_writer_type::skip_all_scales() || writer.skip_track_scale(track_index)))
					return;
				writer.write_scale(track_index, vector4f{ qvv[8], qvv[9], qvv[10], 0.0f });
			}
		}

		device* m_device = nullptr;
		const uint8_t* m_tracks = nullptr;
		aclhip_clip m_clip = ACLHIP_INVALID_HANDLE;
		aclhip_clip_info m_info = {};
		sample_looping_policy m_looping_policy = sample_looping_policy::clamp;
		sample_rounding_policy m_rounding_policy = sample_rounding_policy::none;
		float m_sample_time = -1.0f;
		std::vector<float> m_pose;
		std::vector<float> m_defaults;
		std::vector<uint8_t> m_track_rounding;
	};
}
