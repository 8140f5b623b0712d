// TEST INFRASTRUCTURE — NOT PART OF THE PRODUCT.
//
// Compiles the reference's own, unmodified decompression headers from where they lie under
// /root/reference/includes (nothing is copied) against oracle/rtm_shim/ and exposes the result
// through a tiny C ABI, so that tests, smoke() and bench.py's cpu_baseline leg can run the REAL
// reference CPU decoder on the same compressed bytes. Output goes to oracle/_ref/libaclref.so.
//
// Entry points mirror acl::decompression_context<S>::initialize / seek / decompress_tracks /
// decompress_track (includes/acl/decompression/decompress.h:76-201).
//
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.

#include <acl/core/compressed_tracks.h>
#include <acl/core/track_writer.h>
#include <acl/decompression/decompress.h>
#include <acl/decompression/decompression_settings.h>

#include "bench_harness.h"

#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstring>
#include <memory>
#include <thread>
#include <vector>

#if defined(ACL_ON_ASSERT_THROW)
	#include <exception>
#endif

namespace
{
	// What tools/acl_decompressor/sources/benchmark.cpp:94-101 times: default settings, latest version only,
	// safety checks skipped.
	struct benchmark_settings final : public acl::default_transform_decompression_settings
	{
		static constexpr acl::compressed_tracks_version16 version_supported() { return acl::compressed_tracks_version16::latest; }
		static constexpr bool skip_initialize_safety_checks() { return true; }
	};

	// Every feature on: normalization 'always', per track rounding, all formats, any version.
	using debug_settings = acl::debug_transform_decompression_settings;
	using default_settings = acl::default_transform_decompression_settings;

	// Default settings + per track rounding
	struct per_track_settings final : public acl::default_transform_decompression_settings
	{
		static constexpr bool is_per_track_rounding_supported() { return true; }
	};

	// Default settings that take EVERY packed format (quatf_full / quatf_drop_w_full / vector3f_full next to the variable ones): what a
	// runtime that loads the reference's raw or mixed regression configs (test_data/configs/uniformly_sampled_raw / _mixed_var_*.config.sjson)
	// would compile -- normalization lerp_only, no per track rounding
	struct any_format_settings final : public acl::default_transform_decompression_settings
	{
		static constexpr bool is_rotation_format_supported(acl::rotation_format8) { return true; }
		static constexpr bool is_translation_format_supported(acl::vector_format8) { return true; }
		static constexpr bool is_scale_format_supported(acl::vector_format8) { return true; }
	};

	// ... and the same with rotation_normalization_policy_t::never
	struct any_format_never_normalize_settings final : public acl::default_transform_decompression_settings
	{
		static constexpr bool is_rotation_format_supported(acl::rotation_format8) { return true; }
		static constexpr bool is_translation_format_supported(acl::vector_format8) { return true; }
		static constexpr bool is_scale_format_supported(acl::vector_format8) { return true; }
		static constexpr acl::rotation_normalization_policy_t get_rotation_normalization_policy() { return acl::rotation_normalization_policy_t::never; }
	};

	// 48 bytes per bone: rotation xyzw, translation xyz_, scale xyz_ (core/impl/debug_track_writer.h:61-62,172-192)
	template<acl::default_sub_track_mode rot_mode, acl::default_sub_track_mode trans_mode, acl::default_sub_track_mode scale_mode>
	struct qvv_pose_writer final : public acl::track_writer
	{
		float* out;							// num_tracks * 12 floats
		const float* defaults;				// optional bind pose (num_tracks * 12 floats) for 'variable' mode / constant value (12 floats) for 'constant' mode
		const uint8_t* per_track_policies;	// optional, one sample_rounding_policy per track

		static constexpr acl::default_sub_track_mode get_default_rotation_mode() { return rot_mode; }
		static constexpr acl::default_sub_track_mode get_default_translation_mode() { return trans_mode; }
		static constexpr acl::default_sub_track_mode get_default_scale_mode() { return scale_mode; }

		rtm::quatf RTM_SIMD_CALL get_constant_default_rotation() const { return defaults != nullptr ? rtm::quat_load(defaults + 0) : rtm::quat_identity(); }
		rtm::vector4f RTM_SIMD_CALL get_constant_default_translation() const { return defaults != nullptr ? rtm::vector_load(defaults + 4) : rtm::vector_zero(); }
		rtm::vector4f RTM_SIMD_CALL get_constant_default_scale() const { return defaults != nullptr ? rtm::vector_load(defaults + 8) : rtm::vector_set(1.0F); }

		rtm::quatf RTM_SIMD_CALL get_variable_default_rotation(uint32_t track_index) const { return rtm::quat_load(defaults + track_index * 12 + 0); }
		rtm::vector4f RTM_SIMD_CALL get_variable_default_translation(uint32_t track_index) const { return rtm::vector_load(defaults + track_index * 12 + 4); }
		rtm::vector4f RTM_SIMD_CALL get_variable_default_scale(uint32_t track_index) const { return rtm::vector_load(defaults + track_index * 12 + 8); }

		acl::sample_rounding_policy get_rounding_policy(acl::sample_rounding_policy seek_policy, uint32_t track_index) const
		{
			if (seek_policy == acl::sample_rounding_policy::per_track && per_track_policies != nullptr)
				return static_cast<acl::sample_rounding_policy>(per_track_policies[track_index]);
			return seek_policy;
		}

		void RTM_SIMD_CALL write_rotation(uint32_t track_index, rtm::quatf_arg0 rotation) { rtm::quat_store(rotation, out + track_index * 12 + 0); }
		void RTM_SIMD_CALL write_translation(uint32_t track_index, rtm::vector4f_arg0 translation) { rtm::vector_store3(translation, out + track_index * 12 + 4); }
		void RTM_SIMD_CALL write_scale(uint32_t track_index, rtm::vector4f_arg0 scale) { rtm::vector_store3(scale, out + track_index * 12 + 8); }
	};

	constexpr acl::default_sub_track_mode k_skipped = acl::default_sub_track_mode::skipped;
	constexpr acl::default_sub_track_mode k_constant = acl::default_sub_track_mode::constant;
	constexpr acl::default_sub_track_mode k_variable = acl::default_sub_track_mode::variable;
	constexpr acl::default_sub_track_mode k_legacy = acl::default_sub_track_mode::legacy;

	using writer_identity = qvv_pose_writer<k_constant, k_constant, k_legacy>;		// the track_writer defaults (core/track_writer.h:161-163)
	using writer_skipped = qvv_pose_writer<k_skipped, k_skipped, k_skipped>;
	using writer_constant = qvv_pose_writer<k_constant, k_constant, k_constant>;
	using writer_variable = qvv_pose_writer<k_variable, k_variable, k_variable>;

	template<class settings_t, class writer_t>
	int run_one(const acl::compressed_tracks& tracks, float sample_time, int rounding, int looping, int track_index, writer_t& writer)
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
			context.decompress_track(static_cast<uint32_t>(track_index), writer);
		return 0;
	}

	template<class settings_t>
	int run_one_mode(const acl::compressed_tracks& tracks, float sample_time, int rounding, int looping, int track_index,
		int default_mode, float* out, const float* defaults, const uint8_t* per_track_policies)
	{
		switch (default_mode)
		{
		default:
		case 0: { writer_identity w; w.out = out; w.defaults = nullptr; w.per_track_policies = per_track_policies; return run_one<settings_t>(tracks, sample_time, rounding, looping, track_index, w); }
		case 1: { writer_skipped w; w.out = out; w.defaults = nullptr; w.per_track_policies = per_track_policies; return run_one<settings_t>(tracks, sample_time, rounding, looping, track_index, w); }
		case 2: { writer_constant w; w.out = out; w.defaults = defaults; w.per_track_policies = per_track_policies; return run_one<settings_t>(tracks, sample_time, rounding, looping, track_index, w); }
		case 3: { writer_variable w; w.out = out; w.defaults = defaults; w.per_track_policies = per_track_policies; return run_one<settings_t>(tracks, sample_time, rounding, looping, track_index, w); }
		}
	}
}

extern "C"
{
	// 0 = valid; 1 = is_valid() returned an error (core/impl/compressed_tracks.impl.h:278-301)
	int aclref_is_valid(const void* blob, int check_hash)
	{
		const acl::compressed_tracks* tracks = static_cast<const acl::compressed_tracks*>(blob);
		return tracks->is_valid(check_hash != 0).any() ? 1 : 0;
	}

	// compressed_tracks::get_parent_track_index and get_track_description(track_desc_transformf&) for every track
	// (core/impl/compressed_tracks.impl.h:175-275): returns bit 0 = descriptions are stored. out_parents [n]; out_defaults [n][12]
	// (rotation xyzw | translation xyz 0 | scale xyz 0); out_precisions / out_shell_distances [n]
	int aclref_get_metadata(const void* blob, uint32_t* out_parents, float* out_defaults, float* out_precisions, float* out_shell_distances)
	{
		const acl::compressed_tracks* tracks = static_cast<const acl::compressed_tracks*>(blob);
		int result = 0;
		for (uint32_t track_index = 0; track_index < tracks->get_num_tracks(); ++track_index)
		{
			out_parents[track_index] = tracks->get_parent_track_index(track_index);
			acl::track_desc_transformf desc;
			if (tracks->get_track_description(track_index, desc))
			{
				result = 1;
				float* row = out_defaults + size_t(track_index) * 12;
				std::memset(row, 0, 48);
				rtm::quat_store(desc.default_value.rotation, row + 0);
				rtm::vector_store3(desc.default_value.translation, row + 4);
				rtm::vector_store3(desc.default_value.scale, row + 8);
				out_precisions[track_index] = desc.precision;
				out_shell_distances[track_index] = desc.shell_distance;
				if (desc.parent_index != out_parents[track_index])
					return -1;
			}
		}
		return result;
	}

	// Writes the error string (or "") into 'message'.
	int aclref_is_valid_msg(const void* blob, int check_hash, char* message, int message_capacity)
	{
		const acl::compressed_tracks* tracks = static_cast<const acl::compressed_tracks*>(blob);
		const acl::error_result result = tracks->is_valid(check_hash != 0);
		if (message != nullptr && message_capacity > 0)
		{
			std::strncpy(message, result.c_str(), static_cast<size_t>(message_capacity) - 1);
			message[message_capacity - 1] = '\0';
		}
		return result.any() ? 1 : 0;
	}

	float aclref_get_duration(const void* blob, int looping)
	{
		const acl::compressed_tracks* tracks = static_cast<const acl::compressed_tracks*>(blob);
		return tracks->get_finite_duration(looping < 0 ? acl::sample_looping_policy::as_compressed : static_cast<acl::sample_looping_policy>(looping));
	}

	uint32_t aclref_get_num_tracks(const void* blob) { return static_cast<const acl::compressed_tracks*>(blob)->get_num_tracks(); }
	uint32_t aclref_get_num_samples(const void* blob) { return static_cast<const acl::compressed_tracks*>(blob)->get_num_samples_per_track(); }

	// settings: 0 = default_transform_decompression_settings, 1 = debug_transform_decompression_settings,
	//           2 = default + per track rounding, 3 = benchmark settings (default, latest version only, no safety checks)
	// default_mode: 0 = track_writer defaults (identity rot/trans, legacy scale), 1 = skipped, 2 = constant (defaults[12]), 3 = variable (defaults[num_tracks*12])
	// looping: -1 = as compressed, 0 = clamp, 1 = wrap
	// track_index: -1 = decompress_tracks, otherwise decompress_track(track_index)
	// returns 0 ok, 2 initialize failed, 3 an ACL_ASSERT fired (only in the asserting build)
	int aclref_decompress(const void* blob, float sample_time, int rounding, int looping, int settings, int default_mode,
		int track_index, float* out, const float* defaults, const uint8_t* per_track_policies)
	{
		const acl::compressed_tracks& tracks = *static_cast<const acl::compressed_tracks*>(blob);
#if defined(ACL_ON_ASSERT_THROW)
		try
		{
#endif
			switch (settings)
			{
			default:
			case 0: return run_one_mode<default_settings>(tracks, sample_time, rounding, looping, track_index, default_mode, out, defaults, per_track_policies);
			case 1: return run_one_mode<debug_settings>(tracks, sample_time, rounding, looping, track_index, default_mode, out, defaults, per_track_policies);
			case 2: return run_one_mode<per_track_settings>(tracks, sample_time, rounding, looping, track_index, default_mode, out, defaults, per_track_policies);
			case 3: return run_one_mode<benchmark_settings>(tracks, sample_time, rounding, looping, track_index, default_mode, out, defaults, per_track_policies);
			case 4: return run_one_mode<any_format_settings>(tracks, sample_time, rounding, looping, track_index, default_mode, out, defaults, per_track_policies);
			case 5: return run_one_mode<any_format_never_normalize_settings>(tracks, sample_time, rounding, looping, track_index, default_mode, out, defaults, per_track_policies);
			}
#if defined(ACL_ON_ASSERT_THROW)
		}
		catch (const std::exception& ex)
		{
			std::fprintf(stderr, "aclref: assert: %s\n", ex.what());
			return 3;
		}
#endif
	}

	// Same pose decompressed for 'count' sample times into out[count][num_tracks*12] (used to build golden fixtures).
	int aclref_decompress_many(const void* blob, const float* sample_times, uint32_t count, int rounding, int looping, int settings, int default_mode, float* out)
	{
		const uint32_t num_tracks = aclref_get_num_tracks(blob);
		for (uint32_t i = 0; i < count; ++i)
		{
			const int r = aclref_decompress(blob, sample_times[i], rounding, looping, settings, default_mode, -1, out + size_t(i) * num_tracks * 12, nullptr, nullptr);
			if (r != 0)
				return r;
		}
		return 0;
	}

	// CPU baseline: 'count' instances {clip index, sample time}, statically partitioned over 'num_threads'
	// std::threads (created once), one decompression_context per thread re-initialised when the clip changes, seek +
	// decompress_tracks per instance with the benchmark settings (benchmark.cpp:94-101,249-254). Every thread walks
	// its partition 'repeats' times after one untimed warm-up walk; all threads start the timed part together.
	// If 'out' is null every thread writes into its own private pose buffer (no output kept).
	// Returns the average elapsed seconds of one pass over the whole instance list.
	double aclref_bench(const void* const* blobs, const uint32_t* clip_indices, const float* sample_times, uint32_t count,
		uint32_t max_tracks, uint32_t num_threads, uint32_t repeats, float* out)
	{
		if (num_threads == 0)
			num_threads = 1;
		if (repeats == 0)
			repeats = 1;

		std::atomic<uint32_t> num_ready(0);
		std::atomic<uint32_t> go(0);
		std::atomic<uint32_t> num_done(0);

		auto worker = [&](uint32_t thread_index)
		{
			const uint32_t begin = uint32_t((uint64_t(count) * thread_index) / num_threads);
			const uint32_t end = uint32_t((uint64_t(count) * (thread_index + 1)) / num_threads);

			std::vector<float> scratch(size_t(max_tracks) * 12);
			acl::decompression_context<benchmark_settings> context;
			const void* bound = nullptr;

			for (uint32_t pass = 0; pass <= repeats; ++pass)
			{
				if (pass == 1)
				{
					// warm-up walk done: rendezvous so that the timed passes of all threads overlap
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

					writer_identity writer;
					writer.out = out != nullptr ? out + size_t(i) * max_tracks * 12 : scratch.data();
					writer.defaults = nullptr;
					writer.per_track_policies = nullptr;

					context.seek(sample_times[i], acl::sample_rounding_policy::none);
					context.decompress_tracks(writer);
				}
			}

			num_done.fetch_add(1);
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
		const double elapsed = std::chrono::duration<double>(std::chrono::steady_clock::now() - start).count();

		return elapsed / double(repeats);
	}

	// CPU baseline, self-describing variant (bench.py cpu_baseline leg): the same work per instance as aclref_bench -- one
	// decompression_context per thread re-initialised when the clip changes, seek + decompress_tracks with the benchmark settings
	// (benchmark.cpp:94-101,249-254) into a private pose buffer (warm cache: the favourable case for the CPU) -- but on PINNED
	// threads that start together and run for a fixed wall-clock window (oracle/bench_harness.h). Thread t walks instances
	// [count * t / T, count * (t + 1) / T) again and again. Returns poses per second of all threads together.
	double aclref_bench_timed(const void* const* blobs, const uint32_t* clip_indices, const float* sample_times, uint32_t count,
		uint32_t max_tracks, uint32_t num_threads, double seconds, int pin, uint64_t* out_total_poses)
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
				acl::decompression_context<benchmark_settings> context;
				const void* bound = nullptr;
				uint32_t begin = 0, end = 0, cursor = 0;
			};
			std::shared_ptr<worker_state> state = std::make_shared<worker_state>();
			state->scratch.resize(size_t(max_tracks) * 12);
			state->begin = uint32_t((uint64_t(count) * thread_index) / num_threads);
			state->end = uint32_t((uint64_t(count) * (thread_index + 1)) / num_threads);
			if (state->end == state->begin)		// more threads than instances: share the list
			{
				state->begin = thread_index % count;
				state->end = state->begin + 1;
			}
			state->cursor = state->begin;
			return [state, blobs, clip_indices, sample_times]() -> uint64_t
			{
				worker_state& w = *state;
				constexpr uint32_t k_poses_per_step = 32;
				for (uint32_t k = 0; k < k_poses_per_step; ++k)
				{
					const uint32_t i = w.cursor;
					w.cursor = w.cursor + 1 == w.end ? w.begin : w.cursor + 1;
					const void* blob = blobs[clip_indices[i]];
					if (blob != w.bound)
					{
						w.context.initialize(*static_cast<const acl::compressed_tracks*>(blob));
						w.bound = blob;
					}
					writer_identity writer;
					writer.out = w.scratch.data();
					writer.defaults = nullptr;
					writer.per_track_policies = nullptr;
					w.context.seek(sample_times[i], acl::sample_rounding_policy::none);
					w.context.decompress_tracks(writer);
				}
				return k_poses_per_step;
			};
		};
		return bench_harness::run_timed(num_threads, seconds, pin != 0, make_worker, out_total_poses);
	}

	// The reference's OWN cold-cache protocol (tools/acl_decompressor/sources/benchmark.cpp:232-281), one thread like there: `num_copies`
	// copies of the clip with a context each, a pose decoded from every copy in turn at one sample time, then the CPU caches flushed
	// (a `flush_bytes` buffer rewritten) before the next sample time; only seek + decompress_tracks are timed, the flush is not.
	// What the reference's published numbers are measured with (docs/decompression_performance.md); the warm, pinned sweep above is the
	// figure that favours the CPU. Returns poses per second of that one thread.
	double aclref_bench_cold(const void* blob, uint32_t blob_size, const float* sample_times, uint32_t num_sample_times, uint32_t max_tracks,
		uint32_t num_copies, uint64_t flush_bytes, double seconds)
	{
		if (blob == nullptr || blob_size == 0 || num_sample_times == 0 || num_copies == 0)
			return 0.0;
		std::vector<std::vector<uint8_t>> storage(num_copies);
		std::vector<const acl::compressed_tracks*> copies(num_copies);
		std::vector<acl::decompression_context<benchmark_settings>> contexts(num_copies);
		for (uint32_t c = 0; c < num_copies; ++c)
		{
			storage[c].resize(size_t(blob_size) + 64);
			uint8_t* aligned = reinterpret_cast<uint8_t*>((reinterpret_cast<uintptr_t>(storage[c].data()) + 15) & ~uintptr_t(15));
			std::memcpy(aligned, blob, blob_size);
			copies[c] = reinterpret_cast<const acl::compressed_tracks*>(aligned);
			contexts[c].initialize(*copies[c]);
		}
		std::vector<uint8_t> flush_buffer(flush_bytes);
		std::vector<float> scratch(size_t(max_tracks) * 12);
		uint8_t flush_value = 1;
		std::memset(flush_buffer.data(), flush_value++, flush_buffer.size());
		double timed_seconds = 0.0;
		uint64_t poses = 0;
		uint32_t sample = 0;
		const auto wall_start = std::chrono::steady_clock::now();
		while (std::chrono::duration<double>(std::chrono::steady_clock::now() - wall_start).count() < seconds)
		{
			for (uint32_t c = 0; c < num_copies; ++c)
			{
				const auto start = std::chrono::steady_clock::now();
				writer_identity writer;
				writer.out = scratch.data();
				writer.defaults = nullptr;
				writer.per_track_policies = nullptr;
				contexts[c].seek(sample_times[sample], acl::sample_rounding_policy::none);
				contexts[c].decompress_tracks(writer);
				timed_seconds += std::chrono::duration<double>(std::chrono::steady_clock::now() - start).count();
				poses++;
			}
			sample = sample + 1 == num_sample_times ? 0 : sample + 1;
			std::memset(flush_buffer.data(), flush_value++, flush_buffer.size());
			// (the compiler must not drop the rewrite of a buffer nobody reads)
			asm volatile("" :: "r"(flush_buffer.data()) : "memory");
		}
		return timed_seconds > 0.0 ? double(poses) / timed_seconds : 0.0;
	}
}
