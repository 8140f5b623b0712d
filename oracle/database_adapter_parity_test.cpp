// TEST INFRASTRUCTURE -- not part of the product.
//
// One process, two database-backed decoders behind the SAME interface: acl::database_context + acl::decompression_context (the
// reference's own headers, read in place from /root/reference, compiled against oracle/rtm_shim/, fed by the reference's
// debug_database_streamer) and acl_gpu::database_context + acl_gpu::decompression_context (acl_amd/csrc/acl_gpu_adapter.h over
// libaclhip.so). Both get the same script of stream_in / stream_out requests; after every request the request results and the
// poses of every clip must be identical, bit for bit. Built here into oracle/_ref/database_adapter_parity_test by oracle/Makefile;
// run on the GPU box by tests/test_gpu_adapter.py.
//
// argv: database bulk_medium bulk_low ops clip...      ops: text, one "tier num_chunks stream_in" triple per line
// exit code 0 = all identical, otherwise the number of the failing check
#include <acl/core/ansi_allocator.h>
#include <acl/core/compressed_database.h>
#include <acl/core/compressed_tracks.h>
#include <acl/core/track_writer.h>
#include <acl/decompression/decompress.h>
#include <acl/decompression/database/database.h>
#include <acl/decompression/database/impl/debug_database_streamer.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <vector>

#include "../acl_amd/csrc/acl_gpu_adapter.h"

namespace
{
	struct db_decompression_settings final : public acl::default_transform_decompression_settings
	{
		using database_settings_type = acl::default_database_settings;
	};

	struct pose_writer final : public acl::track_writer
	{
		float* pose = nullptr;
		void RTM_SIMD_CALL write_rotation(uint32_t i, rtm::quatf_arg0 q) { rtm::quat_store(q, pose + size_t(i) * 12); }
		void RTM_SIMD_CALL write_translation(uint32_t i, rtm::vector4f_arg0 v) { rtm::vector_store3(v, pose + size_t(i) * 12 + 4); }
		void RTM_SIMD_CALL write_scale(uint32_t i, rtm::vector4f_arg0 v) { rtm::vector_store3(v, pose + size_t(i) * 12 + 8); }
	};

	struct aligned_file
	{
		std::vector<uint8_t> storage;
		uint8_t* data = nullptr;
		size_t size = 0;

		bool read(const char* path)
		{
			FILE* file = std::fopen(path, "rb");
			if (file == nullptr)
				return false;
			std::fseek(file, 0, SEEK_END);
			size = size_t(std::ftell(file));
			std::fseek(file, 0, SEEK_SET);
			storage.resize(size + 128);
			data = reinterpret_cast<uint8_t*>((reinterpret_cast<uintptr_t>(storage.data()) + 63) & ~uintptr_t(63));
			const bool ok = size == 0 || std::fread(data, 1, size, file) == size;
			std::fclose(file);
			return ok;
		}
	};

	using cpu_context = acl::decompression_context<db_decompression_settings>;
	using gpu_context = acl_gpu::decompression_context<db_decompression_settings>;

	int compare_poses(std::vector<std::unique_ptr<cpu_context>>& cpu, std::vector<std::unique_ptr<gpu_context>>& gpu, const std::vector<aligned_file>& clips)
	{
		const acl::sample_rounding_policy rounding[] = { acl::sample_rounding_policy::none, acl::sample_rounding_policy::floor, acl::sample_rounding_policy::ceil, acl::sample_rounding_policy::nearest };
		for (size_t c = 0; c < clips.size(); ++c)
		{
			const acl::compressed_tracks& tracks = *reinterpret_cast<const acl::compressed_tracks*>(clips[c].data);
			const uint32_t num_tracks = tracks.get_num_tracks();
			const float duration = tracks.get_finite_duration();
			std::vector<float> cpu_pose(size_t(num_tracks) * 12), gpu_pose(size_t(num_tracks) * 12);
			pose_writer cpu_writer, gpu_writer;
			cpu_writer.pose = cpu_pose.data();
			gpu_writer.pose = gpu_pose.data();
			for (const acl::sample_rounding_policy policy : rounding)
			{
				for (int step = 0; step <= 16; ++step)
				{
					const float sample_time = duration * float(step) / 16.0F;
					for (size_t k = 0; k < cpu_pose.size(); ++k)
						cpu_pose[k] = gpu_pose[k] = -7.0F;
					cpu[c]->seek(sample_time, policy);
					gpu[c]->seek(sample_time, policy);
					cpu[c]->decompress_tracks(cpu_writer);
					gpu[c]->decompress_tracks(gpu_writer);
					if (std::memcmp(cpu_pose.data(), gpu_pose.data(), cpu_pose.size() * sizeof(float)) != 0)
					{
						std::fprintf(stderr, "pose mismatch: clip %zu rounding %d time %f\n", c, int(policy), double(sample_time));
						return 1;
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
	if (argc < 6)
		return 100;
	aligned_file database_file, bulk_medium, bulk_low;
	if (!database_file.read(argv[1]) || !bulk_medium.read(argv[2]) || !bulk_low.read(argv[3]))
		return 101;
	std::vector<aligned_file> clips(size_t(argc - 5));
	for (int i = 5; i < argc; ++i)
		if (!clips[size_t(i - 5)].read(argv[i]))
			return 101;

	const acl::compressed_database& database = *reinterpret_cast<const acl::compressed_database*>(database_file.data);
	if (database.is_valid(true).any())
		return 102;

	// ACLHIP_ADAPTER_STREAMERS=1: the GPU context is handed streamer OBJECTS like the reference's (its own pair: a streamer serves one
	// database context), not the bytes they serve
	const char* streamers_mode = std::getenv("ACLHIP_ADAPTER_STREAMERS");
	const bool use_streamer_objects = streamers_mode != nullptr && streamers_mode[0] == '1';

	acl::ansi_allocator allocator;
	acl::debug_database_streamer streamer_medium(allocator, bulk_medium.data, uint32_t(bulk_medium.size));
	acl::debug_database_streamer streamer_low(allocator, bulk_low.data, uint32_t(bulk_low.size));
	acl::debug_database_streamer gpu_streamer_medium(allocator, bulk_medium.data, uint32_t(bulk_medium.size));
	acl::debug_database_streamer gpu_streamer_low(allocator, bulk_low.data, uint32_t(bulk_low.size));
	acl::database_context<acl::default_database_settings> cpu_database;
	acl_gpu::database_context<acl::default_database_settings> gpu_database;
	if (!cpu_database.initialize(allocator, database, streamer_medium, streamer_low))
		return 103;
	if (!(use_streamer_objects ? gpu_database.initialize(allocator, database, gpu_streamer_medium, gpu_streamer_low) : gpu_database.initialize(database, bulk_medium.data, bulk_low.data)))
		return 103;
	if (!gpu_database.is_bound_to(database))
		return 104;

	std::vector<std::unique_ptr<cpu_context>> cpu;
	std::vector<std::unique_ptr<gpu_context>> gpu;
	for (const aligned_file& clip : clips)
	{
		const acl::compressed_tracks& tracks = *reinterpret_cast<const acl::compressed_tracks*>(clip.data);
		if (cpu_database.contains(tracks) != gpu_database.contains(tracks))
			return 105;
		cpu.emplace_back(new cpu_context());
		gpu.emplace_back(new gpu_context());
		if (!cpu.back()->initialize(tracks, cpu_database) || !gpu.back()->initialize(tracks, gpu_database))
			return 106;
	}

	int status = compare_poses(cpu, gpu, clips);
	if (status != 0)
		return 1;

	FILE* ops = std::fopen(argv[4], "r");
	if (ops == nullptr)
		return 101;
	int tier_value = 0, stream_in = 0, num_requests = 0;
	unsigned num_chunks = 0;
	while (std::fscanf(ops, "%d %u %d", &tier_value, &num_chunks, &stream_in) == 3)
	{
		const acl::quality_tier tier = tier_value == 1 ? acl::quality_tier::medium_importance : acl::quality_tier::lowest_importance;
		const acl::database_stream_request_result cpu_result = stream_in ? cpu_database.stream_in(tier, num_chunks) : cpu_database.stream_out(tier, num_chunks);
		const acl::database_stream_request_result gpu_result = stream_in ? gpu_database.stream_in(tier, num_chunks) : gpu_database.stream_out(tier, num_chunks);
		num_requests++;
		if (cpu_result != gpu_result)
		{
			std::fprintf(stderr, "request %d (tier %d, %u chunks, %s): reference says %d, gpu says %d\n", num_requests, tier_value, num_chunks, stream_in ? "in" : "out", int(cpu_result), int(gpu_result));
			return 2;
		}
		for (const acl::quality_tier t : { acl::quality_tier::medium_importance, acl::quality_tier::lowest_importance })
			if (cpu_database.is_streamed_in(t) != gpu_database.is_streamed_in(t))
				return 3;
		status = compare_poses(cpu, gpu, clips);
		if (status != 0)
		{
			std::fprintf(stderr, "after request %d\n", num_requests);
			return 1;
		}
	}
	std::fclose(ops);
	std::printf("%d requests, %zu clips: request results and poses identical\n", num_requests, clips.size());
	return 0;
}
