// TEST INFRASTRUCTURE — NOT PART OF THE PRODUCT.
//
// The reference's own COMPRESSOR (includes/acl/compression/**, unmodified, read in place from /root/reference) compiled
// against oracle/rtm_shim/, exposed through a tiny C ABI. It exists to produce GENUINE compressed_tracks blobs
// (get_default_compression_settings(): variable bit rates, segmenting, constant/default compaction, optional keyframe
// stripping and loop optimisation) from synthetic raw animation, so that fixtures under tests/golden/ are what the reference
// really writes, not what acl_amd/csrc/clip_synth.cpp believes it writes. Output: oracle/_ref/libaclref_compress.so.
#include <acl/core/ansi_allocator.h>
#include <acl/core/compressed_tracks.h>
#include <acl/compression/compress.h>
#include <acl/compression/compression_settings.h>
#include <acl/compression/track_array.h>
#include <acl/compression/transform_error_metrics.h>

#include <cstdint>
#include <cstring>
#include <utility>

extern "C"
{
	// raw: [num_samples][num_tracks][12] floats (rot xyzw | trans xyz_ | scale xyz_). parents: num_tracks entries, -1 = root.
	// flags: bit 0 optimize_loops, bit 1 keyframe stripping (proportion = strip_proportion), bit 2 include metadata (names etc. none; contributing error)
	// Returns the blob size (0 on error). The blob is copied into `out` (16 byte aligned, capacity bytes) when it fits.
	uint32_t aclref_compress(const float* raw, uint32_t num_tracks, uint32_t num_samples, float sample_rate, const int32_t* parents,
		float precision, float shell_distance, uint32_t flags, float strip_proportion, void* out, uint32_t capacity, char* error, uint32_t error_capacity)
	{
		acl::ansi_allocator allocator;
		uint32_t size = 0;
		{
			acl::track_array_qvvf tracks(allocator, num_tracks);
			for (uint32_t track_index = 0; track_index < num_tracks; ++track_index)
			{
				acl::track_desc_transformf desc;
				desc.output_index = track_index;
				desc.parent_index = parents != nullptr && parents[track_index] >= 0 ? uint32_t(parents[track_index]) : acl::k_invalid_track_index;
				desc.precision = precision;
				desc.shell_distance = shell_distance;

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
			settings.optimize_loops = (flags & 1u) != 0;
			if ((flags & 2u) != 0)
			{
				settings.keyframe_stripping.proportion = strip_proportion;
				settings.keyframe_stripping.strip_trivial = true;
			}
			if ((flags & 4u) != 0)
				settings.metadata.include_contributing_error = true;

			acl::output_stats stats;
			acl::compressed_tracks* compressed = nullptr;
			const acl::error_result result = acl::compress_track_list(allocator, tracks, settings, compressed, stats);
			if (result.any() || compressed == nullptr)
			{
				if (error != nullptr && error_capacity != 0)
				{
					std::strncpy(error, result.c_str(), error_capacity - 1);
					error[error_capacity - 1] = '\0';
				}
				return 0;
			}

			size = compressed->get_size();
			if (out != nullptr && capacity >= size)
				std::memcpy(out, compressed, size);
			allocator.deallocate(compressed, size);
		}
		return size;
	}
}
