// TEST INFRASTRUCTURE — NOT PART OF THE PRODUCT.
//
// The reference's database path, unmodified headers read in place from /root/reference, against oracle/rtm_shim/:
//   - build_database / split_database_bulk_data (includes/acl/compression/compress.h:98-124) to produce genuine database bound
//     compressed_tracks, a compressed_database and its bulk data from clips compressed with enable_database_support;
//   - database_context + debug_database_streamer (includes/acl/decompression/database/) to decode at any streaming state.
// Output: oracle/_ref/libaclref_db.so. Used to generate tests/golden/db_*.npz.
#include <acl/core/ansi_allocator.h>
#include <acl/core/compressed_database.h>
#include <acl/core/compressed_tracks.h>
#include <acl/compression/compress.h>
#include <acl/compression/compression_settings.h>
#include <acl/compression/track_array.h>
#include <acl/compression/transform_error_metrics.h>
#include <acl/decompression/decompress.h>
#include <acl/decompression/database/database.h>
#include <acl/decompression/database/impl/debug_database_streamer.h>

#include <cstdint>
#include <cstring>
#include <memory>
#include <utility>
#include <vector>

namespace
{
	struct db_decompression_settings final : public acl::default_transform_decompression_settings
	{
		using database_settings_type = acl::default_database_settings;
	};

	struct qvv_writer final : public acl::track_writer
	{
		float* out;
		void RTM_SIMD_CALL write_rotation(uint32_t track_index, rtm::quatf_arg0 rotation) { rtm::quat_store(rotation, out + track_index * 12 + 0); }
		void RTM_SIMD_CALL write_translation(uint32_t track_index, rtm::vector4f_arg0 translation) { rtm::vector_store3(translation, out + track_index * 12 + 4); }
		void RTM_SIMD_CALL write_scale(uint32_t track_index, rtm::vector4f_arg0 scale) { rtm::vector_store3(scale, out + track_index * 12 + 8); }
	};

	struct built_database
	{
		acl::ansi_allocator allocator;
		std::vector<acl::compressed_tracks*> clips;		// database bound
		acl::compressed_database* database = nullptr;	// bulk data inline
		acl::compressed_database* split = nullptr;		// bulk data split out
		uint8_t* bulk_medium = nullptr;
		uint8_t* bulk_low = nullptr;

		std::unique_ptr<acl::debug_database_streamer> streamer_medium;
		std::unique_ptr<acl::debug_database_streamer> streamer_low;
		std::unique_ptr<acl::database_context<acl::default_database_settings>> context;
	};
}

extern "C"
{
	// Compresses raw animation with database support enabled (contributing error metadata kept), like aclref_compress
	uint32_t aclref_db_compress(const float* raw, uint32_t num_tracks, uint32_t num_samples, float sample_rate, float precision, void* out, uint32_t capacity)
	{
		acl::ansi_allocator allocator;
		uint32_t size = 0;
		{
			acl::track_array_qvvf tracks(allocator, num_tracks);
			for (uint32_t track_index = 0; track_index < num_tracks; ++track_index)
			{
				acl::track_desc_transformf desc;
				desc.output_index = track_index;
				desc.parent_index = track_index == 0 ? acl::k_invalid_track_index : track_index - 1;
				desc.precision = precision;
				desc.shell_distance = 1.0F;
				acl::track_qvvf track = acl::track_qvvf::make_reserve(desc, allocator, num_samples, sample_rate);
				for (uint32_t sample_index = 0; sample_index < num_samples; ++sample_index)
				{
					const float* qvv = raw + (size_t(sample_index) * num_tracks + track_index) * 12;
					track[sample_index] = rtm::qvv_set(rtm::quat_load(qvv + 0), rtm::vector_load3(qvv + 4), rtm::vector_load3(qvv + 8));
				}
				tracks[track_index] = std::move(track);
			}

			acl::compression_settings settings = acl::get_default_compression_settings();
			acl::qvvf_transform_error_metric error_metric;
			settings.error_metric = &error_metric;
			settings.enable_database_support = true;
			settings.keyframe_stripping = acl::compression_keyframe_stripping_settings();	// stripping and database support are mutually exclusive (compression_settings.impl.h:149-150)

			acl::output_stats stats;
			acl::compressed_tracks* compressed = nullptr;
			const acl::error_result result = acl::compress_track_list(allocator, tracks, settings, compressed, stats);
			if (result.any() || compressed == nullptr)
			{
				std::fprintf(stderr, "aclref_db_compress: %s\n", result.c_str());
				return 0;
			}
			size = compressed->get_size();
			if (out != nullptr && capacity >= size)
				std::memcpy(out, compressed, size);
			allocator.deallocate(compressed, size);
		}
		return size;
	}

	void* aclref_db_build(const void* const* clips, uint32_t num_clips, float medium_proportion, float low_proportion, uint32_t max_chunk_size)
	{
		built_database* built = new built_database();
		acl::compression_database_settings settings;
		settings.medium_importance_tier_proportion = medium_proportion;
		settings.low_importance_tier_proportion = low_proportion;
		settings.max_chunk_size = max_chunk_size;

		std::vector<const acl::compressed_tracks*> inputs(num_clips);
		for (uint32_t i = 0; i < num_clips; ++i)
			inputs[i] = static_cast<const acl::compressed_tracks*>(clips[i]);
		built->clips.assign(num_clips, nullptr);

		acl::error_result result = acl::build_database(built->allocator, settings, inputs.data(), num_clips, built->clips.data(), built->database);
		if (result.any())
		{
			std::fprintf(stderr, "aclref_db_build: %s\n", result.c_str());
			delete built;
			return nullptr;
		}

		result = acl::split_database_bulk_data(built->allocator, *built->database, built->split, built->bulk_medium, built->bulk_low);
		if (result.any())
		{
			std::fprintf(stderr, "aclref_db_build (split): %s\n", result.c_str());
			delete built;
			return nullptr;
		}
		return built;
	}

	uint32_t aclref_db_clip_size(void* handle, uint32_t index) { return static_cast<built_database*>(handle)->clips[index]->get_size(); }
	void aclref_db_get_clip(void* handle, uint32_t index, void* out) { built_database* b = static_cast<built_database*>(handle); std::memcpy(out, b->clips[index], b->clips[index]->get_size()); }
	uint32_t aclref_db_database_size(void* handle, int split) { built_database* b = static_cast<built_database*>(handle); return (split ? b->split : b->database)->get_size(); }
	void aclref_db_get_database(void* handle, int split, void* out) { built_database* b = static_cast<built_database*>(handle); const acl::compressed_database* db = split ? b->split : b->database; std::memcpy(out, db, db->get_size()); }
	uint32_t aclref_db_bulk_size(void* handle, int tier) { return static_cast<built_database*>(handle)->split->get_bulk_data_size(tier == 1 ? acl::quality_tier::medium_importance : acl::quality_tier::lowest_importance); }
	void aclref_db_get_bulk(void* handle, int tier, void* out)
	{
		built_database* b = static_cast<built_database*>(handle);
		std::memcpy(out, tier == 1 ? b->bulk_medium : b->bulk_low, aclref_db_bulk_size(handle, tier));
	}
	uint32_t aclref_db_num_chunks(void* handle, int tier) { return static_cast<built_database*>(handle)->split->get_num_chunks(tier == 1 ? acl::quality_tier::medium_importance : acl::quality_tier::lowest_importance); }

	// database_context::initialize(allocator, database, medium_streamer, low_streamer) (database/database.h:116): nothing streamed in yet
	int aclref_db_context_create(void* handle)
	{
		built_database* b = static_cast<built_database*>(handle);
		b->streamer_medium.reset(new acl::debug_database_streamer(b->allocator, b->bulk_medium, aclref_db_bulk_size(handle, 1)));
		b->streamer_low.reset(new acl::debug_database_streamer(b->allocator, b->bulk_low, aclref_db_bulk_size(handle, 2)));
		b->context.reset(new acl::database_context<acl::default_database_settings>());
		return b->context->initialize(b->allocator, *b->split, *b->streamer_medium, *b->streamer_low) ? 0 : 1;
	}

	// database_context::stream_in / stream_out (database/database.h:160-181); the debug streamer completes synchronously
	int aclref_db_stream(void* handle, int tier, uint32_t num_chunks, int stream_in)
	{
		built_database* b = static_cast<built_database*>(handle);
		const acl::quality_tier quality = tier == 1 ? acl::quality_tier::medium_importance : acl::quality_tier::lowest_importance;
		const acl::database_stream_request_result result = stream_in ? b->context->stream_in(quality, num_chunks) : b->context->stream_out(quality, num_chunks);
		return static_cast<int>(result);
	}

	// decompression_context::initialize(tracks, database) + seek + decompress_tracks (decompress.h:108,160,166)
	int aclref_db_decompress(void* handle, uint32_t clip_index, float sample_time, int rounding, float* out)
	{
		built_database* b = static_cast<built_database*>(handle);
		acl::decompression_context<db_decompression_settings> context;
		if (!context.initialize(*b->clips[clip_index], *b->context))
			return 2;
		qvv_writer writer;
		writer.out = out;
		context.seek(sample_time, static_cast<acl::sample_rounding_policy>(rounding));
		context.decompress_tracks(writer);
		return 0;
	}

	// strip_database_quality_tier (compression/compress.h:124): the database without its medium (1) or low (2) importance tier;
	// split != 0 strips the database whose bulk data was split out. Returns the size of the stripped database, 0 on error.
	uint32_t aclref_db_strip(void* handle, int split, int tier, void* out, uint32_t capacity)
	{
		built_database* b = static_cast<built_database*>(handle);
		const acl::compressed_database& source = split ? *b->split : *b->database;
		acl::compressed_database* stripped = nullptr;
		const acl::error_result result = acl::strip_database_quality_tier(b->allocator, source, tier == 1 ? acl::quality_tier::medium_importance : acl::quality_tier::lowest_importance, stripped);
		if (result.any() || stripped == nullptr)
			return 0;
		const uint32_t size = stripped->get_size();
		if (out != nullptr && capacity >= size)
			std::memcpy(out, stripped, size);
		b->allocator.deallocate(stripped, size);
		return size;
	}

	void aclref_db_destroy(void* handle)
	{
		built_database* b = static_cast<built_database*>(handle);
		if (b == nullptr)
			return;
		b->context.reset();
		b->streamer_medium.reset();
		b->streamer_low.reset();
		delete b;	// the ansi_allocator leak check is compiled out without asserts; the process is short lived
	}
}
