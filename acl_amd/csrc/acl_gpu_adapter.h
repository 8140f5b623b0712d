// acl_gpu_adapter.h -- what a maintainer of the reference would add next to includes/acl/decompression/decompress.h:
// acl_gpu::decompression_context<settings> with the interface of acl::decompression_context<settings>
// (/root/reference/includes/acl/decompression/decompress.h:76-201), taking acl::compressed_tracks and acl::track_writer types and
// running on the GPU through aclhip.hpp / libaclhip.so. This header needs the reference's headers (and RTM) on the include path; the
// product itself does not include it. The in-process parity program described in INTEGRATION.md section 2 drives it next to the
// reference's own context.
#pragma once

#include <acl/core/compressed_tracks.h>
#include <acl/core/track_writer.h>
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

	template<class settings_type>
	class decompression_context
	{
	public:
		bool initialize(const acl::compressed_tracks& tracks) { return m_impl.initialize(device(), &tracks, tracks.get_size()); }
		bool is_initialized() const { return m_impl.is_initialized(); }
		void reset() { m_impl.reset(); }
		bool relocated(const acl::compressed_tracks& tracks) { return m_impl.relocated(&tracks); }
		bool is_bound_to(const acl::compressed_tracks& tracks) const { return m_impl.is_bound_to(&tracks); }
		void set_looping_policy(acl::sample_looping_policy policy) { m_impl.set_looping_policy(static_cast<aclhip::sample_looping_policy>(policy)); }
		acl::sample_looping_policy get_looping_policy() const { return static_cast<acl::sample_looping_policy>(m_impl.get_looping_policy()); }
		void seek(float sample_time, acl::sample_rounding_policy policy) { m_impl.seek(sample_time, static_cast<aclhip::sample_rounding_policy>(policy)); }

		template<class writer_type>
		void decompress_tracks(writer_type& writer) { impl::writer_adapter<writer_type> adapter(writer); m_impl.decompress_tracks(adapter); }

		template<class writer_type>
		void decompress_track(uint32_t track_index, writer_type& writer) { impl::writer_adapter<writer_type> adapter(writer); m_impl.decompress_track(track_index, adapter); }

	private:
		aclhip::decompression_context<impl::mapped_settings<settings_type>> m_impl;
	};
}
