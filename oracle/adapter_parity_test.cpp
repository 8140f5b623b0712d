// TEST INFRASTRUCTURE -- not part of the product.
//
// One process, two decoders behind the SAME interface: acl::decompression_context<settings> (the reference's own headers, read in
// place from /root/reference and compiled against oracle/rtm_shim/) and acl_gpu::decompression_context<settings>
// (acl_amd/csrc/acl_gpu_adapter.h over libaclhip.so), driven by the same acl::track_writer types the way the reference's validator
// does (tools/acl_compressor/sources/validate_tracks.cpp:92-260). Every pose must be bit identical. Built here (the reference is not
// on the GPU box) into oracle/_ref/adapter_parity_test by oracle/Makefile; run on the GPU box by tests/test_gpu_adapter.py.
//
// argv: clip file(s)...   exit code 0 = all bit identical, otherwise the number of the failing check
#include <acl/core/compressed_tracks.h>
#include <acl/core/track_writer.h>
#include <acl/decompression/decompress.h>
#include <acl/decompression/decompression_settings.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../acl_amd/csrc/acl_gpu_adapter.h"

namespace
{
	// a writer like core/impl/debug_track_writer.h: rtm::qvvf records, defaults chosen at compile time
	template<acl::default_sub_track_mode mode, acl::default_sub_track_mode scale_mode>
	struct pose_writer final : public acl::track_writer
	{
		float* pose = nullptr;
		const uint8_t* per_track_policies = nullptr;

		static constexpr acl::default_sub_track_mode get_default_rotation_mode() { return mode; }
		static constexpr acl::default_sub_track_mode get_default_translation_mode() { return mode; }
		static constexpr acl::default_sub_track_mode get_default_scale_mode() { return scale_mode; }

		rtm::quatf RTM_SIMD_CALL get_variable_default_rotation(uint32_t i) const { return rtm::quat_set(0.5F, -0.5F, 0.5F, 0.5F + float(i)); }
		rtm::vector4f RTM_SIMD_CALL get_variable_default_translation(uint32_t i) const { return rtm::vector_set(float(i), 2.0F, 3.0F, 0.0F); }
		rtm::vector4f RTM_SIMD_CALL get_variable_default_scale(uint32_t i) const { return rtm::vector_set(2.0F, float(i), 2.0F, 0.0F); }

		acl::sample_rounding_policy get_rounding_policy(acl::sample_rounding_policy policy, uint32_t i) const
		{
			if (policy == acl::sample_rounding_policy::per_track)
				return per_track_policies != nullptr ? static_cast<acl::sample_rounding_policy>(per_track_policies[i]) : acl::sample_rounding_policy::none;
			return policy;
		}

		void RTM_SIMD_CALL write_rotation(uint32_t i, rtm::quatf_arg0 q) { rtm::quat_store(q, pose + size_t(i) * 12); }
		void RTM_SIMD_CALL write_translation(uint32_t i, rtm::vector4f_arg0 v) { rtm::vector_store3(v, pose + size_t(i) * 12 + 4); }
		void RTM_SIMD_CALL write_scale(uint32_t i, rtm::vector4f_arg0 v) { rtm::vector_store3(v, pose + size_t(i) * 12 + 8); }
	};

	// scalar track lists: value[track][component], what write_float1 .. write_vector4 receive
	struct value_writer final : public acl::track_writer
	{
		float* values = nullptr;
		float* pose = nullptr;		// (alias used by compare())
		const uint8_t* per_track_policies = nullptr;
		uint32_t num_components = 1;

		acl::sample_rounding_policy get_rounding_policy(acl::sample_rounding_policy policy, uint32_t i) const
		{
			if (policy == acl::sample_rounding_policy::per_track)
				return per_track_policies != nullptr ? static_cast<acl::sample_rounding_policy>(per_track_policies[i]) : acl::sample_rounding_policy::none;
			return policy;
		}

		void RTM_SIMD_CALL write_float1(uint32_t i, rtm::scalarf_arg0 v) { pose[i] = rtm::scalar_cast(v); }
		void RTM_SIMD_CALL write_float2(uint32_t i, rtm::vector4f_arg0 v) { rtm::vector_store2(v, pose + size_t(i) * 2); }
		void RTM_SIMD_CALL write_float3(uint32_t i, rtm::vector4f_arg0 v) { rtm::vector_store3(v, pose + size_t(i) * 3); }
		void RTM_SIMD_CALL write_float4(uint32_t i, rtm::vector4f_arg0 v) { rtm::vector_store(v, pose + size_t(i) * 4); }
		void RTM_SIMD_CALL write_vector4(uint32_t i, rtm::vector4f_arg0 v) { rtm::vector_store(v, pose + size_t(i) * 4); }
	};

	template<class settings_type, class writer_type>
	int compare(const acl::compressed_tracks& tracks, bool per_track)
	{
		acl::decompression_context<settings_type> cpu;
		acl_gpu::decompression_context<settings_type> gpu;
		if (!cpu.initialize(tracks) || !gpu.initialize(tracks))
			return 1;
		if (!gpu.is_bound_to(tracks) || gpu.get_looping_policy() != cpu.get_looping_policy())
			return 2;

		const uint32_t num_tracks = tracks.get_num_tracks();
		const float duration = tracks.get_finite_duration();
		std::vector<uint8_t> policies(num_tracks);
		for (uint32_t i = 0; i < num_tracks; ++i)
			policies[i] = uint8_t((i * 7 + 1) % 4);

		// 12 floats per track hold a qvv as well as any scalar sample
		std::vector<float> cpu_pose(size_t(num_tracks) * 12 + 4), gpu_pose(size_t(num_tracks) * 12 + 4);
		writer_type cpu_writer, gpu_writer;
		cpu_writer.per_track_policies = gpu_writer.per_track_policies = policies.data();

		const acl::sample_rounding_policy rounding[] = { acl::sample_rounding_policy::none, acl::sample_rounding_policy::floor, acl::sample_rounding_policy::ceil,
			acl::sample_rounding_policy::nearest, acl::sample_rounding_policy::per_track };
		for (int looping = 0; looping < 3; ++looping)
		{
			const acl::sample_looping_policy policy = looping == 0 ? acl::sample_looping_policy::as_compressed : (looping == 1 ? acl::sample_looping_policy::clamp : acl::sample_looping_policy::wrap);
			cpu.set_looping_policy(policy);
			gpu.set_looping_policy(policy);
			for (uint32_t r = 0; r < (per_track ? 5u : 4u); ++r)
			{
				for (int step = -1; step <= 12; ++step)
				{
					const float sample_time = duration * float(step) / 11.0F + 0.0137F * float(step % 3);
					// the same prefill on both sides: skipped default sub-tracks leave it in place
					for (size_t k = 0; k < cpu_pose.size(); ++k)
						cpu_pose[k] = gpu_pose[k] = -3.0F - float(k % 5);
					cpu_writer.pose = cpu_pose.data();
					gpu_writer.pose = gpu_pose.data();
					cpu.seek(sample_time, rounding[r]);
					gpu.seek(sample_time, rounding[r]);
					cpu.decompress_tracks(cpu_writer);
					gpu.decompress_tracks(gpu_writer);
					if (std::memcmp(cpu_pose.data(), gpu_pose.data(), cpu_pose.size() * sizeof(float)) != 0)
					{
						std::fprintf(stderr, "pose mismatch: looping %d rounding %u time %f\n", looping, r, double(sample_time));
						for (size_t k = 0, shown = 0; k < cpu_pose.size() && shown < 8; ++k)
							if (std::memcmp(&cpu_pose[k], &gpu_pose[k], sizeof(float)) != 0)
							{
								std::fprintf(stderr, "  float %zu (track %zu, lane %zu): reference %.9g, gpu %.9g\n", k, k / 12, k % 12, double(cpu_pose[k]), double(gpu_pose[k]));
								shown++;
							}
						return 3;
					}
				}
			}
		}
		return 0;
	}
}

int main(int argc, char** argv)
{
	if (!acl_gpu::device().is_valid())
	{
		std::fprintf(stderr, "no usable HIP device: aclhip_create failed (this program needs a GPU)\n");
		return 99;
	}
	for (int arg = 1; arg < argc; ++arg)
	{
		FILE* file = std::fopen(argv[arg], "rb");
		if (file == nullptr)
			return 100;
		std::fseek(file, 0, SEEK_END);
		const size_t size = size_t(std::ftell(file));
		std::fseek(file, 0, SEEK_SET);
		std::vector<uint8_t> storage(size + 32);
		uint8_t* blob = reinterpret_cast<uint8_t*>((reinterpret_cast<uintptr_t>(storage.data()) + 15) & ~uintptr_t(15));
		if (std::fread(blob, 1, size, file) != size)
			return 101;
		std::fclose(file);

		const acl::compressed_tracks& tracks = *reinterpret_cast<const acl::compressed_tracks*>(blob);
		if (tracks.is_valid(true).any())
			return 102;

		if (tracks.get_track_type() != acl::track_type8::qvvf)
		{
			int result = compare<acl::default_scalar_decompression_settings, value_writer>(tracks, false);
			if (result != 0) return 50 + result;
			result = compare<acl::debug_scalar_decompression_settings, value_writer>(tracks, true);
			if (result != 0) return 60 + result;
			std::printf("%s: bit identical\n", argv[arg]);
			continue;
		}

		using identity_writer = pose_writer<acl::default_sub_track_mode::constant, acl::default_sub_track_mode::legacy>;
		using skipped_writer = pose_writer<acl::default_sub_track_mode::skipped, acl::default_sub_track_mode::skipped>;
		using variable_writer = pose_writer<acl::default_sub_track_mode::variable, acl::default_sub_track_mode::variable>;

		int result = compare<acl::default_transform_decompression_settings, identity_writer>(tracks, false);
		if (result != 0) return 10 + result;
		result = compare<acl::default_transform_decompression_settings, skipped_writer>(tracks, false);
		if (result != 0) return 20 + result;
		result = compare<acl::default_transform_decompression_settings, variable_writer>(tracks, false);
		if (result != 0) return 30 + result;
		result = compare<acl::debug_transform_decompression_settings, identity_writer>(tracks, true);		// always normalize + per track rounding
		if (result != 0) return 40 + result;
		std::printf("%s: bit identical\n", argv[arg]);
	}
	return 0;
}
