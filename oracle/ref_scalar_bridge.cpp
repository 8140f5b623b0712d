// TEST INFRASTRUCTURE -- not part of the product.
//
// The reference's own SCALAR track pipeline (float1f / float2f / float3f / float4f / vector4f track lists), unmodified headers
// read in place from /root/reference and compiled against oracle/rtm_shim/: compress_track_list for raw scalar samples and
// decompression_context<..scalar settings..>::seek / decompress_tracks / decompress_track. Output: oracle/_ref/libaclref_scalar.so.
#include <acl/core/ansi_allocator.h>
#include <acl/core/compressed_tracks.h>
#include <acl/core/track_writer.h>
#include <acl/compression/compress.h>
#include <acl/compression/compression_settings.h>
#include <acl/compression/track_array.h>
#include <acl/decompression/decompress.h>
#include <acl/decompression/decompression_settings.h>

#include "bench_harness.h"

#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstring>
#include <memory>
#include <thread>
#include <utility>
#include <vector>

namespace
{
	using default_settings = acl::default_scalar_decompression_settings;
	using debug_settings = acl::debug_scalar_decompression_settings;		// per track rounding supported

	// out: [num_tracks][num_components] floats, tightly packed
	struct scalar_writer final : public acl::track_writer
	{
		float* out = nullptr;
		uint32_t num_components = 1;
		const uint8_t* per_track_policies = nullptr;

		acl::sample_rounding_policy get_rounding_policy(acl::sample_rounding_policy policy, uint32_t track_index) const
		{
			if (policy == acl::sample_rounding_policy::per_track && per_track_policies != nullptr)
				return static_cast<acl::sample_rounding_policy>(per_track_policies[track_index]);
			return policy == acl::sample_rounding_policy::per_track ? acl::sample_rounding_policy::none : policy;
		}

		void RTM_SIMD_CALL write_float1(uint32_t track_index, rtm::scalarf_arg0 value) { out[track_index] = rtm::scalar_cast(value); }
		void RTM_SIMD_CALL write_float2(uint32_t track_index, rtm::vector4f_arg0 value) { rtm::vector_store2(value, out + size_t(track_index) * 2); }
		void RTM_SIMD_CALL write_float3(uint32_t track_index, rtm::vector4f_arg0 value) { rtm::vector_store3(value, out + size_t(track_index) * 3); }
		void RTM_SIMD_CALL write_float4(uint32_t track_index, rtm::vector4f_arg0 value) { rtm::vector_store(value, out + size_t(track_index) * 4); }
		void RTM_SIMD_CALL write_vector4(uint32_t track_index, rtm::vector4f_arg0 value) { rtm::vector_store(value, out + size_t(track_index) * 4); }
	};

	template<class settings_t>
	int run(const acl::compressed_tracks& tracks, float sample_time, int rounding, int looping, int track_index, scalar_writer& writer)
	{
		acl::decompression_context<settings_t> context;
		if (!context.initialize(tracks))
			return 2;
		if (looping >= 0)
			context.set_looping_policy(static_cast<acl::sample_looping_policy>(looping));
		context.seek(sample_time, static_cast<acl::sample_rounding_policy>(rounding));
		if (track_index < 0)
			context.decompress_tracks(writer);
		else
			context.decompress_track(uint32_t(track_index), writer);
		return 0;
	}

	template<class track_t, class desc_t, class store_t>
	uint32_t compress(uint32_t num_tracks, uint32_t num_samples, float sample_rate, float precision, uint32_t flags, const store_t& store_sample,
		void* out, uint32_t capacity, char* error, uint32_t error_capacity)
	{
		acl::ansi_allocator allocator;
		uint32_t size = 0;
		{
			acl::track_array tracks(allocator, num_tracks);
			for (uint32_t track_index = 0; track_index < num_tracks; ++track_index)
			{
				desc_t desc;
				desc.output_index = track_index;
				desc.precision = precision;
				track_t track = track_t::make_reserve(desc, allocator, num_samples, sample_rate);
				for (uint32_t sample_index = 0; sample_index < num_samples; ++sample_index)
					store_sample(track, track_index, sample_index);
				tracks[track_index] = std::move(track);
			}

			acl::compression_settings settings;		// scalar tracks ignore the transform specific members
			settings.optimize_loops = (flags & 1u) != 0;

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

extern "C"
{
	// track_index < 0: decompress_tracks; settings: 0 = default_scalar_decompression_settings, 1 = debug (per track rounding)
	// looping < 0: keep what the clip says. out: [num_tracks][num_components].
	int aclref_scalar_decompress(const void* blob, float sample_time, int rounding, int looping, int settings, int track_index, float* out,
		const uint8_t* per_track_policies)
	{
		const acl::compressed_tracks& tracks = *static_cast<const acl::compressed_tracks*>(blob);
		scalar_writer writer;
		writer.out = out;
		writer.per_track_policies = per_track_policies;
		return settings == 0 ? run<default_settings>(tracks, sample_time, rounding, looping, track_index, writer)
			: run<debug_settings>(tracks, sample_time, rounding, looping, track_index, writer);
	}

	// raw: [num_samples][num_tracks][num_components] floats. track_type: acl::track_type8 (0 float1f .. 3 float4f, 4 vector4f).
	// flags: bit 0 optimize_loops. Returns the blob size (0 on error); the blob is copied into `out` when it fits.
	uint32_t aclref_scalar_compress(const float* raw, uint32_t track_type, uint32_t num_tracks, uint32_t num_samples, float sample_rate, float precision,
		uint32_t flags, void* out, uint32_t capacity, char* error, uint32_t error_capacity)
	{
		const uint32_t num_components = track_type == 0 ? 1 : (track_type == 1 ? 2 : (track_type == 2 ? 3 : 4));
		const auto sample = [&](uint32_t track_index, uint32_t sample_index) { return raw + (size_t(sample_index) * num_tracks + track_index) * num_components; };
		switch (track_type)
		{
		case 0:
			return compress<acl::track_float1f, acl::track_desc_scalarf>(num_tracks, num_samples, sample_rate, precision, flags,
				[&](acl::track_float1f& track, uint32_t t, uint32_t s) { track[s] = sample(t, s)[0]; }, out, capacity, error, error_capacity);
		case 1:
			return compress<acl::track_float2f, acl::track_desc_scalarf>(num_tracks, num_samples, sample_rate, precision, flags,
				[&](acl::track_float2f& track, uint32_t t, uint32_t s) { track[s] = rtm::float2f{ sample(t, s)[0], sample(t, s)[1] }; }, out, capacity, error, error_capacity);
		case 2:
			return compress<acl::track_float3f, acl::track_desc_scalarf>(num_tracks, num_samples, sample_rate, precision, flags,
				[&](acl::track_float3f& track, uint32_t t, uint32_t s) { track[s] = rtm::float3f{ sample(t, s)[0], sample(t, s)[1], sample(t, s)[2] }; }, out, capacity, error, error_capacity);
		case 3:
			return compress<acl::track_float4f, acl::track_desc_scalarf>(num_tracks, num_samples, sample_rate, precision, flags,
				[&](acl::track_float4f& track, uint32_t t, uint32_t s) { track[s] = rtm::float4f{ sample(t, s)[0], sample(t, s)[1], sample(t, s)[2], sample(t, s)[3] }; }, out, capacity, error, error_capacity);
		case 4:
			return compress<acl::track_vector4f, acl::track_desc_scalarf>(num_tracks, num_samples, sample_rate, precision, flags,
				[&](acl::track_vector4f& track, uint32_t t, uint32_t s) { track[s] = rtm::vector_load(sample(t, s)); }, out, capacity, error, error_capacity);
		default:
			return 0;
		}
	}

	// CPU baseline for scalar track lists, same protocol as aclref_bench (ref_bridge.cpp): 'count' instances statically partitioned
	// over 'num_threads' threads created once, one context per thread, seek + decompress_tracks with the default scalar settings;
	// one untimed warm-up walk, then all threads start the 'repeats' timed walks together. Returns seconds per pass over the list.
	double aclref_scalar_bench(const void* const* blobs, const uint32_t* clip_indices, const float* sample_times, uint32_t count,
		uint32_t max_row_floats, uint32_t num_threads, uint32_t repeats)
	{
		if (num_threads == 0) num_threads = 1;
		if (repeats == 0) repeats = 1;
		std::atomic<uint32_t> num_ready(0);
		std::atomic<uint32_t> go(0);

		auto worker = [&](uint32_t thread_index)
		{
			const uint32_t begin = uint32_t((uint64_t(count) * thread_index) / num_threads);
			const uint32_t end = uint32_t((uint64_t(count) * (thread_index + 1)) / num_threads);
			std::vector<float> scratch(max_row_floats + 4);
			acl::decompression_context<default_settings> context;
			const void* bound = nullptr;
			for (uint32_t pass = 0; pass <= repeats; ++pass)
			{
				if (pass == 1)
				{
					num_ready.fetch_add(1);
					while (go.load() == 0)
						std::this_thread::yield();
				}
				for (uint32_t i = begin; i < end; ++i)
				{
					const void* blob = blobs[clip_indices[i]];
					if (blob != bound)
					{
						context.initialize(*static_cast<const acl::compressed_tracks*>(blob));
						bound = blob;
					}
					scalar_writer writer;
					writer.out = scratch.data();
					context.seek(sample_times[i], acl::sample_rounding_policy::none);
					context.decompress_tracks(writer);
				}
			}
		};

		std::vector<std::thread> threads;
		for (uint32_t t = 0; t < num_threads; ++t)
			threads.emplace_back(worker, t);
		while (num_ready.load() != num_threads)
			std::this_thread::yield();
		const auto start = std::chrono::steady_clock::now();
		go.store(1);
		for (std::thread& t : threads)
			t.join();
		return std::chrono::duration<double>(std::chrono::steady_clock::now() - start).count() / double(repeats);
	}

	// The pinned, time based variant (oracle/bench_harness.h), see aclref_bench_timed in ref_bridge.cpp. Returns track lists per second.
	double aclref_scalar_bench_timed(const void* const* blobs, const uint32_t* clip_indices, const float* sample_times, uint32_t count,
		uint32_t max_row_floats, uint32_t num_threads, double seconds, int pin, uint64_t* out_total)
	{
		if (count == 0)
			return 0.0;
		if (num_threads == 0)
			num_threads = 1;
		auto make_worker = [&](uint32_t thread_index)
		{
			struct worker_state
			{
				std::vector<float> scratch;
				acl::decompression_context<default_settings> context;
				const void* bound = nullptr;
				uint32_t begin = 0, end = 0, cursor = 0;
			};
			std::shared_ptr<worker_state> state = std::make_shared<worker_state>();
			state->scratch.resize(max_row_floats + 4);
			state->begin = uint32_t((uint64_t(count) * thread_index) / num_threads);
			state->end = uint32_t((uint64_t(count) * (thread_index + 1)) / num_threads);
			if (state->end == state->begin)
			{
				state->begin = thread_index % count;
				state->end = state->begin + 1;
			}
			state->cursor = state->begin;
			return [state, blobs, clip_indices, sample_times]() -> uint64_t
			{
				worker_state& w = *state;
				constexpr uint32_t k_lists_per_step = 32;
				for (uint32_t k = 0; k < k_lists_per_step; ++k)
				{
					const uint32_t i = w.cursor;
					w.cursor = w.cursor + 1 == w.end ? w.begin : w.cursor + 1;
					const void* blob = blobs[clip_indices[i]];
					if (blob != w.bound)
					{
						w.context.initialize(*static_cast<const acl::compressed_tracks*>(blob));
						w.bound = blob;
					}
					scalar_writer writer;
					writer.out = w.scratch.data();
					w.context.seek(sample_times[i], acl::sample_rounding_policy::none);
					w.context.decompress_tracks(writer);
				}
				return k_lists_per_step;
			};
		};
		return bench_harness::run_timed(num_threads, seconds, pin != 0, make_worker, out_total);
	}
}
