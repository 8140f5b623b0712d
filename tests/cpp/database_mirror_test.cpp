// Exercises aclhip::database_context + decompression_context::initialize(tracks, database) (acl_amd/csrc/aclhip.hpp) the way
// the reference's database regression drives the real classes (tools/acl_compressor/sources/validate_database.cpp): bind,
// decode with nothing streamed in, stream the medium tier, decode, stream the low tier, decode, stream everything out, decode.
// argv: database bulk_medium bulk_low clip times(text) output(raw floats: 4 states x times x tracks x 12)
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../acl_amd/csrc/aclhip.hpp"

namespace
{
	struct pose_writer : public aclhip::track_writer
	{
		float* pose = nullptr;
		void write_rotation(uint32_t i, aclhip::quatf q) { float* d = pose + size_t(i) * 12; d[0] = q.x; d[1] = q.y; d[2] = q.z; d[3] = q.w; }
		void write_translation(uint32_t i, aclhip::vector4f v) { float* d = pose + size_t(i) * 12 + 4; d[0] = v.x; d[1] = v.y; d[2] = v.z; }
		void write_scale(uint32_t i, aclhip::vector4f v) { float* d = pose + size_t(i) * 12 + 8; d[0] = v.x; d[1] = v.y; d[2] = v.z; }
	};

	// 16 byte aligned copy of a file
	struct buffer
	{
		std::vector<uint8_t> storage;
		uint8_t* data = nullptr;
		size_t size = 0;
	};

	bool read_file(const char* path, buffer& out)
	{
		FILE* file = std::fopen(path, "rb");
		if (file == nullptr)
			return false;
		std::fseek(file, 0, SEEK_END);
		out.size = size_t(std::ftell(file));
		std::fseek(file, 0, SEEK_SET);
		out.storage.resize(out.size + 32);
		out.data = reinterpret_cast<uint8_t*>((reinterpret_cast<uintptr_t>(out.storage.data()) + 15) & ~uintptr_t(15));
		const size_t got = out.size != 0 ? std::fread(out.data, 1, out.size, file) : 0;
		std::fclose(file);
		return got == out.size;
	}
}

int main(int argc, char** argv)
{
	if (argc != 7)
		return 1;
	buffer database, bulk_medium, bulk_low, clip;
	if (!read_file(argv[1], database) || !read_file(argv[2], bulk_medium) || !read_file(argv[3], bulk_low) || !read_file(argv[4], clip))
		return 2;
	std::vector<float> times;
	{
		FILE* file = std::fopen(argv[5], "r");
		if (file == nullptr)
			return 3;
		float t;
		while (std::fscanf(file, "%f", &t) == 1)
			times.push_back(t);
		std::fclose(file);
	}

	using db_context_type = aclhip::database_context<aclhip::default_database_settings>;
	using context_type = aclhip::decompression_context<aclhip::default_transform_decompression_settings>;

	aclhip::device gpu(0);
	if (!gpu.is_valid())
		return 4;

	db_context_type db_context;
	context_type context;

	// an uninitialized database context refuses everything (database.impl.h:447-449, decompress.impl.h:97-99)
	if (db_context.stream_in(aclhip::quality_tier::medium_importance) != aclhip::database_stream_request_result::context_not_initialized)
		return 10;
	if (context.initialize(clip.data, clip.size, db_context))
		return 11;

	if (!db_context.initialize(gpu, database.data, database.size, bulk_medium.size != 0 ? bulk_medium.data : nullptr, bulk_low.size != 0 ? bulk_low.data : nullptr))
		return 12;
	if (!db_context.is_initialized() || !db_context.is_bound_to(database.data) || !db_context.contains(clip.data))
		return 13;
	if (db_context.stream_in(aclhip::quality_tier::highest_importance) != aclhip::database_stream_request_result::invalid_database_tier)
		return 14;
	if (db_context.is_streaming(aclhip::quality_tier::medium_importance))
		return 15;

	if (!context.initialize(clip.data, clip.size, db_context) || !context.is_bound_to(clip.data))
		return 16;

	uint32_t num_tracks;
	std::memcpy(&num_tracks, clip.data + 16, 4);
	std::vector<float> poses(size_t(4) * times.size() * num_tracks * 12, 0.0f);
	pose_writer writer;

	const auto decode_state = [&](uint32_t state)
	{
		for (size_t i = 0; i < times.size(); ++i)
		{
			writer.pose = poses.data() + (size_t(state) * times.size() + i) * num_tracks * 12;
			context.seek(times[i], aclhip::sample_rounding_policy::none);
			context.decompress_tracks(writer);
		}
	};

	decode_state(0);

	const bool has_medium = bulk_medium.size != 0;
	const bool has_low = bulk_low.size != 0;
	if (db_context.is_streamed_in(aclhip::quality_tier::medium_importance) != !has_medium)
		return 20;
	if (db_context.stream_in(aclhip::quality_tier::medium_importance) != (has_medium ? aclhip::database_stream_request_result::dispatched : aclhip::database_stream_request_result::done))
		return 21;
	if (!db_context.is_streamed_in(aclhip::quality_tier::medium_importance))
		return 22;
	if (db_context.stream_in(aclhip::quality_tier::medium_importance) != aclhip::database_stream_request_result::done)
		return 23;
	decode_state(1);

	if (db_context.stream_in(aclhip::quality_tier::lowest_importance) != (has_low ? aclhip::database_stream_request_result::dispatched : aclhip::database_stream_request_result::done))
		return 24;
	decode_state(2);

	db_context.stream_out(aclhip::quality_tier::medium_importance);
	db_context.stream_out(aclhip::quality_tier::lowest_importance);
	if (has_low && db_context.is_streamed_in(aclhip::quality_tier::lowest_importance))
		return 25;
	decode_state(3);

	// the database stays registered while a decompression context is bound to it
	db_context.reset();
	if (!db_context.is_initialized())
		return 30;
	context.reset();
	db_context.reset();
	if (db_context.is_initialized())
		return 31;

	FILE* out = std::fopen(argv[6], "wb");
	if (out == nullptr)
		return 5;
	std::fwrite(poses.data(), sizeof(float), poses.size(), out);
	std::fclose(out);
	return 0;
}
