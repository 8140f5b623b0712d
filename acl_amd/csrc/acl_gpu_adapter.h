// acl_gpu_adapter.h -- what a maintainer of the reference would add next to includes/acl/decompression/decompress.h:
// acl_gpu::decompression_context<settings> with the interface of acl::decompression_context<settings>
// (/root/reference/includes/acl/decompression/decompress.h:76-201), taking acl::compressed_tracks and acl::track_writer types and
// running on the GPU through aclhip.hpp / libaclhip.so. This header needs the reference's headers (and RTM) on the include path; the
// product itself does not include it. The in-process parity program described in INTEGRATION.md section 2 drives it next to the
// reference's own context.
#pragma once

#include <acl/core/compressed_database.h>
#include <acl/core/compressed_tracks.h>
#include <acl/core/track_writer.h>
#include <acl/decompression/database/database.h>
#include <acl/decompression/database/database_streamer.h>
#include <acl/decompression/decompression_settings.h>

#include "aclhip.hpp"

namespace acl_gpu
{
	inline aclhip::device& device() { static aclhip::device gpu(0); return gpu; }

	namespace impl
	{
		// acl settings -> aclhip settings with the same switches
		template<class acl_settings_type>
		struct mapped_settings : public aclhip::decompression_settings
		{
			static constexpr bool clamp_sample_time() { return acl_settings_type::clamp_sample_time(); }
			static constexpr bool is_wrapping_supported() { return acl_settings_type::is_wrapping_supported(); }
			static constexpr bool is_per_track_rounding_supported() { return acl_settings_type::is_per_track_rounding_supported(); }
			static constexpr aclhip::rotation_normalization_policy_t get_rotation_normalization_policy()
			{
				return static_cast<aclhip::rotation_normalization_policy_t>(acl_settings_type::get_rotation_normalization_policy());
			}
		};

		// forwards the aclhip::track_writer protocol to an acl::track_writer
		template<class writer_type>
		struct writer_adapter : public aclhip::track_writer
		{
			writer_type& writer;
			explicit writer_adapter(writer_type& writer_) : writer(writer_) {}

			static constexpr aclhip::default_sub_track_mode convert(acl::default_sub_track_mode mode)
			{
				return mode == acl::default_sub_track_mode::skipped ? aclhip::default_sub_track_mode::skipped
					: mode == acl::default_sub_track_mode::constant ? aclhip::default_sub_track_mode::constant
					: mode == acl::default_sub_track_mode::variable ? aclhip::default_sub_track_mode::variable : aclhip::default_sub_track_mode::legacy;
			}
			static constexpr aclhip::default_sub_track_mode get_default_rotation_mode() { return convert(writer_type::get_default_rotation_mode()); }
			static constexpr aclhip::default_sub_track_mode get_default_translation_mode() { return convert(writer_type::get_default_translation_mode()); }
			static constexpr aclhip::default_sub_track_mode get_default_scale_mode() { return convert(writer_type::get_default_scale_mode()); }

			static aclhip::quatf to_quat(rtm::quatf_arg0 q) { float v[4]; rtm::quat_store(q, v); return aclhip::quatf{ v[0], v[1], v[2], v[3] }; }
			static aclhip::vector4f to_vector(rtm::vector4f_arg0 v_) { float v[4]; rtm::vector_store(v_, v); return aclhip::vector4f{ v[0], v[1], v[2], v[3] }; }

			aclhip::quatf get_constant_default_rotation() const { return to_quat(writer.get_constant_default_rotation()); }
			aclhip::vector4f get_constant_default_translation() const { return to_vector(writer.get_constant_default_translation()); }
			aclhip::vector4f get_constant_default_scale() const { return to_vector(writer.get_constant_default_scale()); }
			aclhip::quatf get_variable_default_rotation(uint32_t i) const { return to_quat(writer.get_variable_default_rotation(i)); }
			aclhip::vector4f get_variable_default_translation(uint32_t i) const { return to_vector(writer.get_variable_default_translation(i)); }
			aclhip::vector4f get_variable_default_scale(uint32_t i) const { return to_vector(writer.get_variable_default_scale(i)); }

			aclhip::sample_rounding_policy get_rounding_policy(aclhip::sample_rounding_policy policy, uint32_t i) const
			{
				return static_cast<aclhip::sample_rounding_policy>(writer.get_rounding_policy(static_cast<acl::sample_rounding_policy>(policy), i));
			}

			bool skip_track_rotation(uint32_t i) const { return writer.skip_track_rotation(i); }
			bool skip_track_translation(uint32_t i) const { return writer.skip_track_translation(i); }
			bool skip_track_scale(uint32_t i) const { return writer.skip_track_scale(i); }
			void write_rotation(uint32_t i, aclhip::quatf q) { writer.write_rotation(i, rtm::quat_set(q.x, q.y, q.z, q.w)); }
			void write_translation(uint32_t i, aclhip::vector4f v) { writer.write_translation(i, rtm::vector_set(v.x, v.y, v.z, 0.0F)); }
			void write_scale(uint32_t i, aclhip::vector4f v) { writer.write_scale(i, rtm::vector_set(v.x, v.y, v.z, 0.0F)); }

			bool skip_track_float1(uint32_t i) const { return writer.skip_track_float1(i); }
			bool skip_track_float2(uint32_t i) const { return writer.skip_track_float2(i); }
			bool skip_track_float3(uint32_t i) const { return writer.skip_track_float3(i); }
			bool skip_track_float4(uint32_t i) const { return writer.skip_track_float4(i); }
			bool skip_track_vector4(uint32_t i) const { return writer.skip_track_vector4(i); }
			void write_float1(uint32_t i, float v) { writer.write_float1(i, rtm::scalar_set(v)); }
			void write_float2(uint32_t i, aclhip::vector4f v) { writer.write_float2(i, rtm::vector_set(v.x, v.y, v.z, v.w)); }
			void write_float3(uint32_t i, aclhip::vector4f v) { writer.write_float3(i, rtm::vector_set(v.x, v.y, v.z, v.w)); }
			void write_float4(uint32_t i, aclhip::vector4f v) { writer.write_float4(i, rtm::vector_set(v.x, v.y, v.z, v.w)); }
			void write_vector4(uint32_t i, aclhip::vector4f v) { writer.write_vector4(i, rtm::vector_set(v.x, v.y, v.z, v.w)); }
		};
	}

	namespace impl
	{
		// what a decompression context needs of the database it is bound to, whatever its settings type
		struct database_context_base
		{
			virtual ~database_context_base() = default;
			virtual void update() const = 0;		// bring the GPU side up to date with requests the caller's streamers have completed
		};
	}

	// acl::database_context<settings> (includes/acl/decompression/database/database.h:69-201) on the GPU, in two forms:
	//  * initialize(database [, bulk_data_medium, bulk_data_low]): the GPU side owns the residency. What it takes is the bytes the
	//    reference's streamers would serve (what debug_database_streamer is constructed with); a streaming request is a stream ordered
	//    copy + metadata update, nothing is ever "in progress" for the caller to poll.
	//  * initialize(allocator, database, medium_streamer, low_streamer) -- the reference's own signature (database.h:116): the caller's
	//    acl::database_streamer objects serve the bulk data, synchronously or not. They are driven by the reference's own
	//    database_context, kept inside (chunk selection, request ids, complete() / cancel(), database_streamer.impl.h:60-172: a streamer
	//    only befriends that class); every request it reports complete is mirrored on the GPU: the range the streamer filled is copied
	//    to HBM and the tier metadata published (aclhip_database_stream_in_from / _stream_out). Requests of an asynchronous streamer
	//    are mirrored when they have completed -- at the latest by the next query, request or decode.
	template<class database_settings_type>
	class database_context final : public impl::database_context_base
	{
	public:
		// reference: initialize(allocator, database) -- bulk data inline (database.h:110)
		bool initialize(const acl::compressed_database& database) { m_streamers[0] = m_streamers[1] = nullptr; return m_impl.initialize(device(), &database, database.get_size()); }
		// the bytes the reference's streamers would serve
		bool initialize(const acl::compressed_database& database, const uint8_t* bulk_data_medium, const uint8_t* bulk_data_low)
		{
			m_streamers[0] = m_streamers[1] = nullptr;
			return m_impl.initialize(device(), &database, database.get_size(), bulk_data_medium, bulk_data_low);
		}
		// reference: initialize(allocator, database, medium_tier_streamer, low_tier_streamer) (database.h:116)
		bool initialize(acl::iallocator& allocator, const acl::compressed_database& database, acl::database_streamer& medium_tier_streamer, acl::database_streamer& low_tier_streamer)
		{
			reset();
			if (!m_reference.initialize(allocator, database, medium_tier_streamer, low_tier_streamer))
				return false;
			if (!m_impl.initialize_streamed(device(), &database, database.get_size()))
			{
				m_reference.reset();
				return false;
			}
			m_streamers[0] = &medium_tier_streamer;
			m_streamers[1] = &low_tier_streamer;
			return true;
		}
		bool is_initialized() const { return m_impl.is_initialized(); }
		void reset()
		{
			m_impl.reset();
			if (m_reference.is_initialized())
				m_reference.reset();
			m_streamers[0] = m_streamers[1] = nullptr;
			m_num_pending = 0;
		}
		bool is_bound_to(const acl::compressed_database& database) const { return m_impl.is_bound_to(&database); }
		bool contains(const acl::compressed_tracks& tracks) const { return m_impl.contains(&tracks); }
		bool is_streamed_in(acl::quality_tier tier) const { update(); return m_impl.is_streamed_in(static_cast<aclhip::quality_tier>(tier)); }
		bool is_streaming(acl::quality_tier tier) const
		{
			update();
			for (uint32_t i = 0; i < m_num_pending; ++i)
				if (m_pending[i].tier == tier)
					return true;
			return m_impl.is_streaming(static_cast<aclhip::quality_tier>(tier));
		}
		acl::database_stream_request_result stream_in(acl::quality_tier tier, uint32_t num_chunks_to_stream = ~0U) { return request(tier, num_chunks_to_stream, true); }
		acl::database_stream_request_result stream_out(acl::quality_tier tier, uint32_t num_chunks_to_stream = ~0U) { return request(tier, num_chunks_to_stream, false); }

		const aclhip::database_context<aclhip::default_database_settings>& get() const { return m_impl; }

		// Mirrors the requests the caller's streamers have completed since the last call (in request order; a request that is still
		// in flight holds back the later ones of its tier)
		void update() const override
		{
			uint32_t kept = 0;
			for (uint32_t i = 0; i < m_num_pending; ++i)
			{
				const pending_request request = m_pending[i];
				bool blocked = m_reference.is_streaming(request.tier);
				for (uint32_t k = 0; k < kept && !blocked; ++k)
					blocked = m_pending[k].tier == request.tier;
				if (blocked)
				{
					m_pending[kept++] = request;
					continue;
				}
				const aclhip::quality_tier tier = static_cast<aclhip::quality_tier>(request.tier);
				if (request.stream_in)
					(void)m_impl.stream_in_from(tier, request.num_chunks, m_streamers[uint32_t(request.tier) - 1]->get_bulk_data(request.tier));
				else
					(void)m_impl.stream_out(tier, request.num_chunks);
			}
			m_num_pending = kept;
		}

	private:
		static acl::database_stream_request_result convert(aclhip::database_stream_request_result result)
		{
			switch (result)
			{
			case aclhip::database_stream_request_result::done:						return acl::database_stream_request_result::done;
			case aclhip::database_stream_request_result::dispatched:				return acl::database_stream_request_result::dispatched;
			case aclhip::database_stream_request_result::streaming_in_progress:		return acl::database_stream_request_result::streaming_in_progress;
			case aclhip::database_stream_request_result::context_not_initialized:	return acl::database_stream_request_result::context_not_initialized;
			case aclhip::database_stream_request_result::invalid_database_tier:		return acl::database_stream_request_result::invalid_database_tier;
			default:																return acl::database_stream_request_result::no_free_streaming_requests;
			}
		}

		acl::database_stream_request_result request(acl::quality_tier tier, uint32_t num_chunks_to_stream, bool stream_in)
		{
			if (m_streamers[0] == nullptr)
				return convert(stream_in ? m_impl.stream_in(static_cast<aclhip::quality_tier>(tier), num_chunks_to_stream) : m_impl.stream_out(static_cast<aclhip::quality_tier>(tier), num_chunks_to_stream));

			// the caller's streamers: the reference's context decides and drives them, the GPU side follows
			update();
			if (m_num_pending == k_max_pending)
				return acl::database_stream_request_result::no_free_streaming_requests;
			const acl::database_stream_request_result result = stream_in ? m_reference.stream_in(tier, num_chunks_to_stream) : m_reference.stream_out(tier, num_chunks_to_stream);
			if (result == acl::database_stream_request_result::dispatched)
			{
				m_pending[m_num_pending++] = pending_request{ tier, num_chunks_to_stream, stream_in };
				update();		// a synchronous streamer has completed already
			}
			return result;
		}

		struct pending_request { acl::quality_tier tier; uint32_t num_chunks; bool stream_in; };
		static constexpr uint32_t k_max_pending = 8;

		mutable aclhip::database_context<aclhip::default_database_settings> m_impl;
		acl::database_context<database_settings_type> m_reference;		// bookkeeping for the caller's streamers only: nothing decodes through it
		acl::database_streamer* m_streamers[2] = { nullptr, nullptr };
		mutable pending_request m_pending[k_max_pending] = {};
		mutable uint32_t m_num_pending = 0;
	};

	template<class settings_type>
	class decompression_context
	{
	public:
		bool initialize(const acl::compressed_tracks& tracks) { return m_impl.initialize(device(), &tracks, tracks.get_size()); }
		// reference: initialize(tracks, database_context) (decompress.h:108): false when the database does not contain the clip
		template<class database_settings_type>
		bool initialize(const acl::compressed_tracks& tracks, const database_context<database_settings_type>& database)
		{
			m_database = &database;
			return m_impl.initialize(&tracks, tracks.get_size(), database.get());
		}
		bool is_initialized() const { return m_impl.is_initialized(); }
		void reset() { m_impl.reset(); m_database = nullptr; }
		bool relocated(const acl::compressed_tracks& tracks) { return m_impl.relocated(&tracks); }
		bool is_bound_to(const acl::compressed_tracks& tracks) const { return m_impl.is_bound_to(&tracks); }
		void set_looping_policy(acl::sample_looping_policy policy) { m_impl.set_looping_policy(static_cast<aclhip::sample_looping_policy>(policy)); }
		acl::sample_looping_policy get_looping_policy() const { return static_cast<acl::sample_looping_policy>(m_impl.get_looping_policy()); }
		void seek(float sample_time, acl::sample_rounding_policy policy) { m_impl.seek(sample_time, static_cast<aclhip::sample_rounding_policy>(policy)); }

		template<class writer_type>
		void decompress_tracks(writer_type& writer) { follow_database(); impl::writer_adapter<writer_type> adapter(writer); m_impl.decompress_tracks(adapter); }

		template<class writer_type>
		void decompress_track(uint32_t track_index, writer_type& writer) { follow_database(); impl::writer_adapter<writer_type> adapter(writer); m_impl.decompress_track(track_index, adapter); }

	private:
		// requests the caller's streamers completed since the last decode become visible first
		void follow_database() const { if (m_database != nullptr) m_database->update(); }

		aclhip::decompression_context<impl::mapped_settings<settings_type>> m_impl;
		const impl::database_context_base* m_database = nullptr;
	};
}
