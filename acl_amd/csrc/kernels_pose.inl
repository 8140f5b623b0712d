// kernels_pose.inl -- part of aclhip.hip (one translation unit; included there, in this order, not compiled on its own).
// The pose kernels: decompress_tracks_kernel, decompress_tracks_any_settings_kernel (one wave64 per instance and pose window).

	constexpr uint32_t k_wave_size = 64;

	// Measurement aid (-DACLHIP_EXP_PHASE_TIMES, tools/phase_times.py): wall clock stamps of the phases of a workgroup (pose consumers) / of a workgroup's first wave (pose kernel)
#if defined(ACLHIP_EXP_PHASE_TIMES)
	__device__ unsigned long long phase_times[16384 * 4];
#define ACLHIP_PHASE_STAMP(k) do { if ((threadIdx.x & 63) == 0 && blockIdx.x < 16384) phase_times[blockIdx.x * 4 + (k)] = wall_clock64(); } while (0)
#else
#define ACLHIP_PHASE_STAMP(k) do { } while (0)
#endif
#if defined(ACLHIP_EXP_PHASE_TIMES) && ACLHIP_EXP_PHASE_TIMES == 2
// = 2: the steps of the pose kernel's PROLOGUE instead (entry, inputs arrived, clip record arrived, seek finished)
#define ACLHIP_WAVE0_STAMP(k) do { if ((k) == 0 && threadIdx.x == 0 && blockIdx.x < 16384) phase_times[blockIdx.x * 4] = wall_clock64(); } while (0)
#define ACLHIP_PROLOGUE_STAMP(k) do { if (threadIdx.x == 0 && blockIdx.x < 16384) phase_times[blockIdx.x * 4 + (k)] = wall_clock64(); } while (0)
#elif defined(ACLHIP_EXP_PHASE_TIMES)
#define ACLHIP_WAVE0_STAMP(k) do { if (threadIdx.x == 0 && blockIdx.x < 16384) phase_times[blockIdx.x * 4 + (k)] = wall_clock64(); } while (0)
#define ACLHIP_PROLOGUE_STAMP(k) do { } while (0)
#else
#define ACLHIP_WAVE0_STAMP(k) do { } while (0)
#define ACLHIP_PROLOGUE_STAMP(k) do { } while (0)
#endif

#if !defined(ACLHIP_WAVES_PER_BLOCK)
	#define ACLHIP_WAVES_PER_BLOCK 4
#endif
	constexpr uint32_t k_waves_per_block = ACLHIP_WAVES_PER_BLOCK;
	constexpr uint32_t k_block_size = k_wave_size * k_waves_per_block;

	// Value of a default sub-track (unpack_default_*_sub_tracks, decompression.transform.h:575-675,883-985,1203-1310, and the
	// "no scale" loop :1653-1680). `identity` is the track_writer default for the kind (identity / zero / legacy scale).
	__device__ __forceinline__ float4 default_quad(const decode_params& params, uint32_t kind, uint32_t track_index, float4 identity, bool& out_store, const float* clip_bind_pose)
	{
		const uint32_t mode = params.default_modes[kind];
		out_store = mode != ACLHIP_DEFAULT_SKIPPED;

		if (mode == ACLHIP_DEFAULT_BIND_POSE)
		{
			// the clip's own table (aclhip_device.h: bind_pose_of)
			const float* src = clip_bind_pose + size_t(track_index) * 12 + kind * 4;
			return make_float4(src[0], src[1], src[2], kind == 0 ? src[3] : 0.0f);
		}
		if (params.default_values != nullptr && (mode == ACLHIP_DEFAULT_CONSTANT || mode == ACLHIP_DEFAULT_VARIABLE))
		{
			const float* src = params.default_values + (mode == ACLHIP_DEFAULT_VARIABLE ? size_t(track_index) * 12 : 0) + kind * 4;
			return make_float4(src[0], src[1], src[2], kind == 0 ? src[3] : 0.0f);
		}

		if (kind == 2 && mode != ACLHIP_DEFAULT_LEGACY)
			return make_float4(1.0f, 1.0f, 1.0f, 0.0f);		// track_writer::get_constant_default_scale (core/track_writer.h:169)

		return identity;
	}

	typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
	typedef uint32_t u32x16 __attribute__((ext_vector_type(16)));
	typedef float f32x4 __attribute__((ext_vector_type(4)));

	// The whole 128 byte clip record in two scalar loads (wave uniform address)
	__device__ __forceinline__ device_clip load_clip(const device_clip* clips, uint32_t clip_id)
	{
		const ACLHIP_CONSTANT u32x16* source = (const ACLHIP_CONSTANT u32x16*)(clips + clip_id);
		struct { u32x16 lo, hi; } raw = { source[0], source[1] };
		// (both halves are requested HERE, together: left alone the compiler sinks the load of the half that only the decode needs below
		// the launch shape checks on the other -- one more dependent round trip in every wave's scalar prologue)
		asm volatile("" : "+s"(raw.lo), "+s"(raw.hi));
		device_clip clip;
		__builtin_memcpy(&clip, &raw, sizeof(clip));
		return clip;
	}

	// The same record with every field in registers of its own. load_clip hands back two 16 register blocks and the fields stay
	// sub-registers of them: a block lives as long as ANY of its fields, and when the scalar registers run out it is spilled and
	// reloaded whole. Kernels that hold two or three records (the pose consumers: clip, base, blend partners) pass theirs through here
	// -- the copies are free where the allocator can leave a field where it was loaded.
	__device__ __forceinline__ device_clip load_clip_fields(const device_clip* clips, uint32_t clip_id)
	{
		device_clip clip = load_clip(clips, clip_id);
		asm volatile("" : "+s"(clip.blob), "+s"(clip.base_pose), "+s"(clip.samples), "+s"(clip.resolved_pose), "+s"(clip.plan), "+s"(clip.clip_ranges), "+s"(clip.image_chunks), "+s"(clip.hierarchy));
		asm volatile("" : "+s"(clip.db_headers), "+s"(clip.db_bulk_data[0]), "+s"(clip.db_bulk_data[1]), "+s"(clip.db_clip_header_offset));
		asm volatile("" : "+s"(clip.num_tracks), "+s"(clip.num_samples), "+s"(clip.sample_rate), "+s"(clip.duration_clamp), "+s"(clip.duration_wrap), "+s"(clip.flags), "+s"(clip.num_segments), "+s"(clip.num_animated));
		return clip;
	}

	// A 32 byte table entry (plan_entry / clip_range_entry) in two 16 byte loads
	template<class entry_t>
	__device__ __forceinline__ entry_t load_entry(const entry_t* table, uint32_t index)
	{
		static_assert(sizeof(entry_t) == 32, "two dwordx4 loads");
		const ACLHIP_CONSTANT u32x4* source = (const ACLHIP_CONSTANT u32x4*)(table + index);
		struct { u32x4 lo, hi; } raw = { source[0], source[1] };
		entry_t entry;
		__builtin_memcpy(&entry, &raw, sizeof(entry));
		return entry;
	}

	__device__ __forceinline__ float4 load_quad(const float4* table, uint32_t index)
	{
		const f32x4 raw = ((const ACLHIP_CONSTANT f32x4*)table)[index];
		return make_float4(raw.x, raw.y, raw.z, raw.w);
	}

	// A launch's shape -- waves per instance, LDS quads per wave -- and the caller's pose stride are decided when the launch is ENQUEUED
	// (pose_launch_shape_of, host_launch.inl), the clip an instance names is read when the wave RUNS: a captured hipGraph replayed after
	// a larger clip was registered, or a caller whose stride is too small for a clip of its batch, would otherwise decode a pose into
	// too few windows (tracks left stale without a word), into an LDS slot that overlaps the next wave's, or past the end of its row.
	// Wave uniform, a handful of scalar instructions: such an instance is refused and counted like an unknown handle -- the reference's
	// silent return (decompression.transform.h:1532-1537) -- and its pose row keeps what the caller had there.
	__device__ __forceinline__ uint32_t layout_bytes_per_track(uint32_t layout)
	{
		return layout == ACLHIP_LAYOUT_QVV48 ? 48u : (layout == ACLHIP_LAYOUT_QVV40 ? 40u : 32u);
	}

	// num_tracks: the tracks of the clip this instance stores -- all of them, or its first K (aclhip_output_desc::instance_track_counts)
	__device__ __forceinline__ bool launch_refuses_clip(uint32_t num_tracks, uint32_t windows_per_instance, uint32_t lds_quads_per_wave, uint32_t bytes_per_track, uint64_t pose_stride_bytes)
	{
		return num_pose_windows(num_tracks) > windows_per_instance
			|| min(num_tracks * 3u, k_image_chunk_quads) > lds_quads_per_wave
			|| uint64_t(num_tracks) * bytes_per_track > pose_stride_bytes;
	}

	// the tracks of its clip an instance stores: aclhip_output_desc::instance_track_counts (wave uniform)
	__device__ __forceinline__ uint32_t stored_tracks_of(const decode_params& params, uint32_t caller_instance, uint32_t clip_tracks)
	{
		return params.instance_track_counts != nullptr ? min(clip_tracks, as_constant(params.instance_track_counts)[caller_instance]) : clip_tracks;
	}

	// What the any-settings pose kernel stores for a quad of the LDS image: default sub-tracks -- still tagged in their W lane, every
	// other quad holds a real W >= +0 by now -- follow the default sub-track modes, the rest passes through.
	__device__ __forceinline__ float4 resolve_quad(const decode_params& params, float4 value, uint32_t quad, bool& out_store, const float* clip_bind_pose)
	{
		out_store = true;
		const uint32_t marker = __float_as_uint(value.w);
		if (!is_special_quad(marker))
			return value;

		const uint32_t track_index = quad / 3u;
		const uint32_t kind = quad - track_index * 3u;
		value.w = (marker & k_quad_default_w_one) != 0 ? 1.0f : 0.0f;
		return default_quad(params, kind, track_index, value, out_store, clip_bind_pose);
	}

	// Where the tables of a pose window's animated sub-tracks are
	struct window_tables
	{
		const plan_entry* plan;					// [num_segments][num_animated]
		const clip_range_entry* clip_ranges;	// [num_animated]
		uint32_t num_animated;
		bool short_exact_math;					// k_clip_short_exact_math of the clip (wave uniform)
	};

	__device__ __forceinline__ window_tables window_tables_of(const device_clip& clip)
	{
		window_tables tables;
		tables.plan = clip.plan;
		tables.clip_ranges = clip.clip_ranges;
		tables.num_animated = clip.num_animated;
		tables.short_exact_math = (clip.flags & k_clip_short_exact_math) != 0;
		return tables;
	}

	// Lanes <-> the animated sub-tracks [first_ordinal, end_ordinal) of one pose window, decoded into their quads of the window's LDS
	// image (image[0] = quad first_quad of the pose), 64 per pass. A pass:
	//   1. every lane fetches its 32 byte plan entry (two when the keys straddle two segments -- most sample times fall between two
	//      keyframes of ONE segment and its row is fetched once: a third less table traffic through the texture unit) and its clip range;
	//   2. the keyframe bits of both keys (four loads in flight), unpack, ranges, W, lerp, normalize.
	// Where a decoded sub-track goes in the window's LDS image: the QVV48 image holds the pose's quads as they are
	struct qvv48_image_writer
	{
		f32x4* image;
		uint32_t first_quad;
		uint32_t window_quads;		// (an instance that stores its first K tracks only has a shorter window than its clip: what lies beyond is dropped)
		__device__ __forceinline__ void operator()(const clip_range_entry& entry, float4 value) const
		{
			if (entry.quad_index - first_quad < window_quads)
				image[entry.quad_index - first_quad] = f32x4{ value.x, value.y, value.z, value.w };
		}
	};

	// What an instance brings to the launch: its clip handle and its sample time (wave uniform, scalar unit). A batch's input lists
	// are read once, by one wave each: these two loads are the ones of a wave's prologue that miss the caches. Both are requested
	// TOGETHER -- left alone the compiler requests the sample time where the seek first uses it, behind the clip record: two misses in a
	// row in every wave's life (found in the ISA in round 5; the phase stamps, profiles/r05_experiments.md 8, put the prologue's steps at
	// 0.3 us for the inputs, 0.25 us for the clip record and 1.7 us from there to the end of the seek -- most of that last figure was the
	// wait for the base pose DMA that uniform_instance_byte, aclhip_device.h, removes). Together with that and the clip range requested
	// in front of the plan entries: -1.3 .. -2.8 % on every kernel of one-window poses (exp_r5t.sh), nothing on the others.
	//   Instance lists decode in slot order: the caller's index of a slot's instance is in their order (decode_params::time_indices),
	//   the sample time that was requested for the slot's own index is thrown away (the index is in bounds: the order is a permutation);
	//   an attached list's clip handles are in the caller's order as well (one more dependent load, for these launches only).
	struct instance_inputs
	{
		uint32_t clip_id;
		uint32_t caller_instance;
		float sample_time;
	};

	// time_with_the_handle = false (known when the kernel is compiled): the sample time where the compiler puts it -- the kernel of poses
	// of several windows, bound by instruction issue, not by the lives of its waves, loses 1.2 % to the three changes together (exp_r5u.sh)
	__device__ __forceinline__ void load_instance_inputs(const uint32_t* __restrict__ clip_ids, const float* __restrict__ sample_times, const decode_params& params,
		uint32_t instance, uint64_t pose_stride_bytes, uint32_t lds_quads_per_wave, bool time_with_the_handle, instance_inputs& out)
	{
		uint32_t clip_id = as_constant(clip_ids)[instance];
		float sample_time = 0.0f;
		// (kernel arguments beyond the 16 preloaded SGPRs -- the stride and the LDS slot size the launch shape checks need -- are fetched
		// next to the instance's clip handle, not behind the clip record where the compiler would put them: one round trip less)
		if (time_with_the_handle)
		{
			sample_time = as_constant(sample_times)[instance];
			asm volatile("" :: "s"(pose_stride_bytes), "s"(lds_quads_per_wave), "s"(clip_id), "s"(__float_as_uint(sample_time)));
		}
		else
			asm volatile("" :: "s"(pose_stride_bytes), "s"(lds_quads_per_wave), "s"(clip_id));
		uint32_t caller_instance = instance;
		if (params.time_indices != nullptr)
		{
			caller_instance = as_constant(params.time_indices)[instance];
			if (params.clips_by_caller_instance != 0)
				clip_id = as_constant(clip_ids)[caller_instance];
			if (time_with_the_handle)
				sample_time = as_constant(sample_times)[caller_instance];
		}
		if (!time_with_the_handle)
			sample_time = as_constant(sample_times)[caller_instance];
		out.clip_id = clip_id;
		out.caller_instance = caller_instance;
		out.sample_time = sample_time;
	}

	template<bool kPolicies, bool kWideKeyLoads = false, uint32_t kFastMath = 0, class image_writer_type>
	__device__ __forceinline__ void decode_window_sub_tracks_into(const window_tables& tables, const seek_state& state, const decode_params& params,
		uint32_t rounding_policy, uint32_t normalization, uint32_t first_ordinal, uint32_t end_ordinal, uint32_t lane, image_writer_type write_to_image,
		const uint8_t* track_rounding_policies = nullptr, bool clip_range_first = false)
	{
		// track_rounding_policies (kPolicies): the writer's per track policies -- the launch's, or the instance's own table (wave uniform)
		// clip_range_first (known when the kernel is compiled): the first pass's clip range is requested in front of its plan entries
		(void)params;
		if (first_ordinal >= end_ordinal)
			return;

		// kPolicies: per track rounding (and the sample normalization it implies)
		const bool normalize_samples = kPolicies && normalization == ACLHIP_NORMALIZE_ALWAYS;
		const bool single_segment = state.uses_single_segment;		// wave uniform
		const plan_entry* plan_row0 = tables.plan + size_t(state.segment_index[0]) * tables.num_animated;
		const plan_entry* plan_row1 = tables.plan + size_t(state.segment_index[1]) * tables.num_animated;

		uint32_t ordinal = min(first_ordinal + lane, end_ordinal - 1);
		// (the clip range first, in the kernels of one-window poses: the keys' addresses wait for the plan entries, and what is requested
		// behind that wait arrives a round trip later)
		clip_range_entry clip_range;
		if (clip_range_first)
			clip_range = load_entry(tables.clip_ranges, ordinal);
		plan_entry entry0 = load_entry(plan_row0, ordinal);
		plan_entry entry1 = single_segment ? entry0 : load_entry(plan_row1, ordinal);
		if (!clip_range_first)
			clip_range = load_entry(tables.clip_ranges, ordinal);

		for (uint32_t base = first_ordinal; base < end_ordinal; base += k_wave_size)
		{
			const bool valid = base + lane < end_ordinal;
			const plan_entry plan0 = entry0, plan1 = entry1;
			const clip_range_entry current_range = clip_range;

			const bool is_rotation = is_rotation_entry(current_range);

			uint32_t policy = k_round_none;
			if (kPolicies)
			{
				// track_writer::get_rounding_policy (core/track_writer.h:97)
				policy = rounding_policy;
				if (rounding_policy == k_round_per_track)
					policy = track_rounding_policies != nullptr ? track_rounding_policies[current_range.track_index] : k_round_none;
			}

			// the raw bit rate is rare: only a wave that actually meets one (in these two segments) pays for its code path
			const bool has_raw = __any(int(is_raw_width(plan0.bit_offset_and_width >> 24) || is_raw_width(plan1.bit_offset_and_width >> 24))) != 0;

			float4 value;
			if (!has_raw)
				value = decode_animated_sub_track<false, kPolicies, kWideKeyLoads, kFastMath>(state, plan0, plan1, current_range, is_rotation, policy, state.interpolation_alpha, normalization, normalize_samples, tables.short_exact_math);
			else
				value = decode_animated_sub_track<true, kPolicies, kWideKeyLoads, (kFastMath == 2 ? 0u : kFastMath)>(state, plan0, plan1, current_range, is_rotation, policy, state.interpolation_alpha, normalization, normalize_samples, tables.short_exact_math);		// (ACLHIP_DECODE_FAST: a wave that meets a raw sample keeps the exact forms)

			// a decoded W is never negative (a square root, or +0): the marker the base pose carried in this quad is gone
			if (valid)
				write_to_image(current_range, value);

			// (requesting the next pass's entries BEFORE this pass's arithmetic was measured: 24 more live registers spill, 2 x slower)
			if (base + k_wave_size < end_ordinal)
			{
				ordinal = min(base + k_wave_size + lane, end_ordinal - 1);
				entry0 = load_entry(plan_row0, ordinal);
				entry1 = single_segment ? entry0 : load_entry(plan_row1, ordinal);
				clip_range = load_entry(tables.clip_ranges, ordinal);
			}
		}
	}

	template<bool kAnySettings, bool kWideKeyLoads = false, uint32_t kFastMath = 0>
	__device__ __forceinline__ void decode_window_sub_tracks(const window_tables& tables, const seek_state& state, const decode_params& params,
		uint32_t rounding_policy, uint32_t normalization, uint32_t first_ordinal, uint32_t end_ordinal, uint32_t first_quad, uint32_t window_quads, uint32_t lane, f32x4* image,
		const uint8_t* track_rounding_policies = nullptr, bool clip_range_first = false)
	{
		const qvv48_image_writer writer = { image, first_quad, window_quads };
		if (kAnySettings && params.per_track_rounding != 0)
			decode_window_sub_tracks_into<true, kWideKeyLoads>(tables, state, params, rounding_policy, normalization, first_ordinal, end_ordinal, lane, writer, track_rounding_policies, clip_range_first);
		else
			decode_window_sub_tracks_into<false, kWideKeyLoads, kFastMath>(tables, state, params, rounding_policy, normalization, first_ordinal, end_ordinal, lane, writer, nullptr, clip_range_first);
	}

	// The pose kernels. One wave64 per (instance, pose window): a window is k_image_chunk_quads consecutive quads of the pose (a
	// 100 bone pose is a single window), built in 5 KiB of LDS:
	//   1. the scalar prologue finds the clip and seeks (4 dependent scalar loads);
	//   2. meanwhile the window's slice of the clip's base pose is DMA'd global -> LDS (global_load_lds, no VGPRs);
	//   3. lanes <-> the animated sub-tracks that land in the window (a contiguous range of ordinals: the tables are ordered by
	//      window) decode straight into their quad of the LDS image;
	//   4. the finished window streams out, 16 bytes per lane, 1 KiB of contiguous HBM per store instruction.
	// Windows of one pose go to consecutive waves: each repeats the (scalar) seek, none waits for another, and the chain of
	// dependent memory round trips per wave stays as short as for a small pose.
	//
	// kAnySettings = false is the common case -- track_writer defaults, no per track rounding, normalization != always: the DMA source
	// is the clip's RESOLVED pose (defaults written out) and step 4 is a plain copy. kAnySettings = true takes every settings
	// combination: the DMA source is the marker tagged base pose, the decode honours per track rounding, and step 4 resolves what
	// is not animated (default sub-track modes, caller supplied defaults, always-normalize).
	// kFastMath = 2: ACLHIP_DECODE_FAST (aclhip_decompress_params::flags) -- rotations through v_sqrt_f32 / v_rsq_f32 and fused
	// multiply-adds (within 2e-6 of the default's; x, y, z of every sample, translations and scales bit identical), for the launches
	// whose time is VALU issue: poses of several windows
	// The pose kernels' arguments as they lie in the kernarg segment (ACLHIP_POSE_KERNEL_ARGUMENTS below: in order, each at its natural
	// alignment). The kernels that take work items in turn read their arguments THERE, per item and where they are used, instead of
	// holding them in scalar registers across the decode: their launch does not fit the 94 SGPRs of 7 waves per SIMD, and what the
	// compiler spills it reloads with v_readlane -- vector instructions, ~30 per window on a kernel bound by vector issue (round 6).
	struct pose_kernel_args
	{
		const device_clip* clips;
		uint32_t num_clips;
		const uint32_t* clip_ids;
		const float* sample_times;
		uint32_t num_instances;
		uint32_t windows_per_instance;
		decode_params params;
		uint8_t* poses;
		uint64_t pose_stride_bytes;
		uint32_t lds_quads_per_wave;
		unsigned long long* rejected_count;
	};

	template<bool kAnySettings, bool kCompactOutput, bool kWideKeyLoads = false, uint32_t kFastMath = 0>
	__device__ __forceinline__ void decompress_tracks_window(const device_clip* __restrict__ clips, uint32_t num_clips,
		const uint32_t* __restrict__ clip_ids, const float* __restrict__ sample_times, uint32_t num_instances, uint32_t windows_per_instance,
		const decode_params& params, uint8_t* __restrict__ poses, uint64_t pose_stride_bytes, uint32_t lds_quads_per_wave,
		unsigned long long* __restrict__ rejected_count, uint32_t work_item, uint32_t* image_clip = nullptr, const ACLHIP_CONSTANT pose_kernel_args* late_args = nullptr)
	{
		extern __shared__ __attribute__((aligned(16))) uint8_t dynamic_lds[];
		ACLHIP_WAVE0_STAMP(0);

		const uint32_t lane = threadIdx.x & (k_wave_size - 1);
		const uint32_t wave_in_block = __builtin_amdgcn_readfirstlane(threadIdx.x / k_wave_size);
		uint32_t instance = work_item;
		uint32_t window = 0;
		if (windows_per_instance != 1)
		{
			instance = work_item / windows_per_instance;
			window = work_item - instance * windows_per_instance;
		}
		if (instance >= num_instances)
			return;

		// wave uniform prologue on the scalar unit: instance -> clip record -> sample records
		// (items taken in turn -- the kernels of poses of several windows -- keep the prologue they had: see load_instance_inputs)
		const bool one_shot = image_clip == nullptr;		// known when the kernel is compiled
		instance_inputs inputs;
		load_instance_inputs(clip_ids, sample_times, params, instance, pose_stride_bytes, lds_quads_per_wave, one_shot, inputs);
		const uint32_t clip_id = inputs.clip_id, caller_instance = inputs.caller_instance;
		const float sample_time = inputs.sample_time;
		ACLHIP_PROLOGUE_STAMP(1);		// (-DACLHIP_EXP_PHASE_TIMES=2: the prologue's steps instead of the wave's phases)
		// (the instance's own policies, if the launch has any, are requested with the clip record: uniform_instance_byte, aclhip_device.h)
		// (the generic compact kernels hold more of the launch in scalar registers: there the two are requested in front of the seek,
		// still on the scalar unit -- requested here they cost those kernels 30 spilled SGPRs)
		uint32_t rounding_policy = 0, looping_policy = 0;
		const bool policies_with_the_clip_record = !kCompactOutput && one_shot;
		if (policies_with_the_clip_record)
		{
			rounding_policy = uniform_instance_rounding_policy_of(params, caller_instance, clips);
			looping_policy = uniform_instance_looping_policy_of(params, caller_instance, clips);
		}
		const device_clip clip = load_clip(clips, clip_id < num_clips ? clip_id : 0);
#if defined(ACLHIP_EXP_PHASE_TIMES) && ACLHIP_EXP_PHASE_TIMES == 2
		asm volatile("" :: "s"(clip.flags));
		ACLHIP_PROLOGUE_STAMP(2);
#endif
		// the tracks this instance stores: all of its clip's, or its first K (aclhip_output_desc::instance_track_counts: a per character LOD)
		const uint32_t stored_tracks = stored_tracks_of(params, caller_instance, clip.num_tracks);
		if (clip_id >= num_clips || !is_transform_clip(clip.flags)
			|| launch_refuses_clip(stored_tracks, windows_per_instance, lds_quads_per_wave, kCompactOutput ? layout_bytes_per_track(params.layout) : 48u, pose_stride_bytes))
		{
			if (lane == 0 && window == 0)
				atomicAdd(late_args != nullptr ? late_args->rejected_count : rejected_count, 1ull);
			return;
		}

		// an empty track list (decompression.transform.h:1531-1533) or a pose that ends before this window
		const uint32_t num_quads = stored_tracks * 3u;
		const uint32_t first_quad = window * k_image_chunk_quads;
		if (first_quad >= num_quads)
			return;
		const uint32_t window_quads = min(num_quads - first_quad, k_image_chunk_quads);

		// the window's animated sub-tracks: image_chunks[window] .. image_chunks[window + 1]
		uint32_t first_ordinal = 0, end_ordinal = clip.num_animated;
		if (clip.num_tracks * 3u > k_image_chunk_quads)
		{
			first_ordinal = as_constant(clip.image_chunks)[window];
			end_ordinal = as_constant(clip.image_chunks)[window + 1];
		}

		f32x4* image = reinterpret_cast<f32x4*>(dynamic_lds) + size_t(wave_in_block) * lds_quads_per_wave;

		// With the track_writer's own default sub-track modes the resolved pose already holds what default sub-tracks decode to; any
		// other mode starts from the tagged base pose and resolves the tags when the window is stored
		const bool resolve_defaults = kAnySettings && params.standard_default_modes == 0;

		// base pose window -> LDS image, asynchronously: lane i of pass p fetches quad first + p * 64 + i into image[p * 64 + i]
		{
			const ACLHIP_CONSTANT f32x4* source = (const ACLHIP_CONSTANT f32x4*)(resolve_defaults ? clip.base_pose : clip.resolved_pose) + first_quad;
			// (in-turn kernel: the image still holds this clip's window from the wave's previous item -- every animated quad is about to be
			// overwritten, the others are this clip's constants already)
			// (with per instance track counts the image of the previous item may hold fewer quads than this one needs: no reuse)
			const bool image_is_current = image_clip != nullptr && *image_clip == clip_id && params.instance_track_counts == nullptr;
			if (image_clip != nullptr)
				*image_clip = clip_id;
			if (!image_is_current)
			{
			for (uint32_t base = 0; base < window_quads; base += k_wave_size)
			{
				if (base + lane < window_quads)
					__builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(source + base + lane),
						(__attribute__((address_space(3))) void*)(image + base), 16, 0, 0);
			}
			}
		}

		const uint32_t normalization = params.normalization;
		if (!policies_with_the_clip_record && one_shot)
		{
			rounding_policy = uniform_instance_rounding_policy_of(params, caller_instance, clips);
			looping_policy = uniform_instance_looping_policy_of(params, caller_instance, clips);
		}
		if (!one_shot)
		{
			rounding_policy = __builtin_amdgcn_readfirstlane(instance_rounding_policy_of(params, caller_instance));
			looping_policy = __builtin_amdgcn_readfirstlane(instance_looping_policy_of(params, caller_instance));
		}

		seek_state state;
		seek(clip, sample_time, rounding_policy, looping_policy, state);
#if defined(ACLHIP_EXP_PHASE_TIMES)
		asm volatile("" :: "s"(state.key_frame_bit_offsets[0]), "s"(state.key_frame_bit_offsets[1]));		// the seek's loads have arrived
		ACLHIP_WAVE0_STAMP(1);
		ACLHIP_PROLOGUE_STAMP(3);
#endif

		if (kAnySettings && normalization == ACLHIP_NORMALIZE_ALWAYS && (clip.flags & k_clip_full_rotations) == 0)
		{
			// rotation_normalization_policy_t::always also normalizes CONSTANT rotations (constant_track_cache.transform.h:163-175):
			// done in the image before the animated sub-tracks (normalized by their decode) replace their markers. Not for quatf_full
			// clips, whose constants are stored whole and left alone (:136-149). Rare (debug
			// settings): this path waits for the base pose instead of overlapping it with the decode.
			__builtin_amdgcn_s_waitcnt(0);
			__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
			__builtin_amdgcn_wave_barrier();
			__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
			for (uint32_t quad = lane; quad < window_quads; quad += k_wave_size)
			{
				// (slots of animated rotations hold a tag, or zeros in the resolved pose: whatever this makes of them is overwritten)
				const f32x4 value = image[quad];
				if ((first_quad + quad) % 3u == 0 && !is_special_quad(__float_as_uint(value.w)))
				{
					const float4 normalized = quat_normalize(make_float4(value.x, value.y, value.z, value.w));
					image[quad] = f32x4{ normalized.x, normalized.y, normalized.z, normalized.w };
				}
			}
		}

		// lanes <-> animated sub-tracks of this window
		// (decoded quads are written after the DMA has delivered their slots: a wave's memory operations return in order, and the
		// keyframe loads every decoded value waits for were issued after the DMA)
		// track_writer::get_rounding_policy per track: the launch's table, or the instance's own writer's (wave uniform)
		const uint8_t* track_rounding_policies = kAnySettings ? params.track_rounding_policies : nullptr;
		if (kAnySettings && params.instance_rounding_tables != nullptr)
			track_rounding_policies = params.track_rounding_table + size_t(as_constant(params.instance_rounding_tables)[caller_instance]) * params.track_rounding_stride;
		decode_window_sub_tracks<kAnySettings, kWideKeyLoads, kFastMath>(window_tables_of(clip), state, params, rounding_policy, normalization, first_ordinal, end_ordinal, first_quad, window_quads, lane, image, track_rounding_policies, one_shot);

		// DMA and the wave's own LDS writes must have landed before lanes read each other's quads
		__builtin_amdgcn_s_waitcnt(0);
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
		__builtin_amdgcn_wave_barrier();
		__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
		ACLHIP_WAVE0_STAMP(2);

		// track_writer::skip_track_rotation / _translation / _scale(track_index) (core/track_writer.h:189-191): one mask for the launch, or
		// this instance's own out of the caller's table (aclhip_output_desc::mask_table, instance_masks)
		const uint8_t* skip_tracks = params.skip_tracks;
		if (kCompactOutput && params.instance_masks != nullptr)
			skip_tracks = params.mask_table + size_t(as_constant(params.instance_masks)[caller_instance]) * params.mask_stride;

		if (kCompactOutput && params.layout != ACLHIP_LAYOUT_QVV48 && (params.skip_mask & ~(params.layout == ACLHIP_LAYOUT_QV32 ? 4u : 0u)) == 0 && !resolve_defaults && skip_tracks == nullptr)
		{
			// Compact layouts, nothing else skipped (the common use): the window is RE-TILED on its way out -- lanes <-> consecutive 16 byte
			// pieces of the OUTPUT, gathered from the QVV48 image in LDS -- so that every store instruction still writes 1 KiB of
			// contiguous HBM. Windows hold whole tracks (k_image_chunk_quads is a multiple of 3) and an even number of them, so a window's
			// output starts 16 byte aligned in both layouts. (Storing per image quad instead leaves holes between the lanes of a store:
			// 76 us instead of 33 us for the QV32 headline batch, partial cache lines at both ends of every instruction.)
			const uint32_t row = params.instance_rows != nullptr ? as_constant(params.instance_rows)[instance] : instance;
			const uint32_t first_track = first_quad / 3u;
			const uint32_t window_tracks = window_quads / 3u;
			if (params.layout == ACLHIP_LAYOUT_QV32)
			{
				f32x4* out = reinterpret_cast<f32x4*>(poses + uint64_t(row) * pose_stride_bytes + size_t(first_track) * 32);
				const uint32_t out_quads = window_tracks * 2u;
				constexpr uint32_t k_out_rows = (k_image_chunk_quads / 3u * 2u + k_wave_size - 1) / k_wave_size;
				f32x4 staged_out[k_out_rows];
				#pragma unroll
				for (uint32_t r = 0; r < k_out_rows; ++r)
				{
					const uint32_t piece = min(r * k_wave_size + lane, out_quads - 1);
					staged_out[r] = image[(piece >> 1) * 3u + (piece & 1u)];
				}
				#pragma unroll
				for (uint32_t r = 0; r < k_out_rows; ++r)
					if (r * k_wave_size + lane < out_quads)
						store_streaming(&out[r * k_wave_size + lane], staged_out[r]);
			}
			else
			{
				// QVV40: float f of a track's 10 comes from float f + (f >= 7) of its QVV48 record (the padding lane of the translation is dropped)
				float* out = reinterpret_cast<float*>(poses + uint64_t(row) * pose_stride_bytes + size_t(first_track) * 40);
				const float* image_floats = reinterpret_cast<const float*>(image);
				const uint32_t out_floats = window_tracks * 10u;
				constexpr uint32_t k_out_rows = (k_image_chunk_quads / 3u * 10u / 4u + k_wave_size) / k_wave_size;
				#pragma unroll
				for (uint32_t r = 0; r < k_out_rows; ++r)
				{
					const uint32_t first_float = (r * k_wave_size + lane) * 4u;
					float piece[4];
					#pragma unroll
					for (uint32_t j = 0; j < 4; ++j)
					{
						const uint32_t f = min(first_float + j, out_floats - 1);
						const uint32_t track = f / 10u;
						const uint32_t component = f - track * 10u;
						piece[j] = image_floats[track * 12u + component + (component >= 7u ? 1u : 0u)];
					}
					if (first_float + 4u <= out_floats)
						store_streaming_floats<4>(out + first_float, piece);
					else if (first_float + 2u <= out_floats)
					{
						// an odd number of tracks ends on half a piece
						const float half[2] = { piece[0], piece[1] };
						store_streaming_floats<2>(out + first_float, half);
					}
				}
			}
			return;
		}

		// LDS -> registers -> HBM: the whole window is read first, then the stores go out back to back from one base address with
		// immediate offsets; full 1 KiB rows take no per lane predicate, only the last (partial) row does
		constexpr uint32_t k_rows = (k_image_chunk_quads + k_wave_size - 1) / k_wave_size;
		const uint32_t full_rows = window_quads / k_wave_size;			// wave uniform
		f32x4 staged[k_rows];
		#pragma unroll
		for (uint32_t r = 0; r < k_rows; ++r)
			staged[r] = image[min(r * k_wave_size + lane, lds_quads_per_wave - 1)];

		// (the row is read here, not in the prologue: one SGPR pair less across the decode; items taken in turn read where their pose goes
		// from the kernarg segment here, see pose_kernel_args)
		const uint32_t* instance_rows = params.instance_rows;
#if defined(__HIP_DEVICE_COMPILE__)
		if (late_args != nullptr)
		{
			const ACLHIP_CONSTANT pose_kernel_args* args = late_args;
			asm volatile("" : "+s"(args));
			instance_rows = args->params.instance_rows;
			poses = args->poses;
			pose_stride_bytes = args->pose_stride_bytes;
		}
#endif
		const uint32_t row = instance_rows != nullptr ? as_constant(instance_rows)[instance] : instance;
		uint8_t* pose_bytes = poses + uint64_t(row) * pose_stride_bytes;
		f32x4* pose = reinterpret_cast<f32x4*>(pose_bytes) + first_quad + lane;

		// any-settings: rows are 64 quads apart and 64 % 3 == 1, so a lane's sub-track kind advances by one per row
		const uint32_t lane_quad = first_quad + lane;
		const uint32_t lane_track = lane_quad / 3u;
		uint32_t kind = lane_quad - lane_track * 3u;
		// (ACLHIP_DEFAULT_BIND_POSE reads the clip's own table: params.user_defaults is set for it as well, resolve_params)
		const bool user_defaults = kAnySettings && params.user_defaults != 0;		// wave uniform

		#pragma unroll
		for (uint32_t r = 0; r < k_rows; ++r)
		{
			bool store = r < full_rows || (r == full_rows && r * k_wave_size + lane < window_quads);
			f32x4 value = staged[r];
			if (kAnySettings && resolve_defaults)
			{
				// default sub-tracks still carry their tag in the W lane (every other quad holds a real W >= +0 by now) and follow the
				// default sub-track modes (unpack_default_*_sub_tracks, decompression.transform.h:575-675,883-985,1203-1310,1653-1680)
				const uint32_t marker = __float_as_uint(value.w);
				const bool is_default = is_special_quad(marker);
				const uint32_t mode = kind == 0 ? params.default_modes[0] : (kind == 1 ? params.default_modes[1] : params.default_modes[2]);
				store = store && !(is_default && mode == ACLHIP_DEFAULT_SKIPPED);
				if (is_default)
				{
					// the image holds the track_writer default's xyz (identity / zero / the clip's legacy default scale)
					value.w = (marker & k_quad_default_w_one) != 0 ? 1.0f : 0.0f;
					if (kind == 2 && mode != ACLHIP_DEFAULT_LEGACY)
						value = f32x4{ 1.0f, 1.0f, 1.0f, 0.0f };		// track_writer::get_constant_default_scale (core/track_writer.h:169)
				}
				if (user_defaults && is_default && ((params.default_values != nullptr && (mode == ACLHIP_DEFAULT_CONSTANT || mode == ACLHIP_DEFAULT_VARIABLE)) || mode == ACLHIP_DEFAULT_BIND_POSE))
				{
					const uint32_t track_index = (lane_quad + r * k_wave_size) / 3u;
					const float* table = mode == ACLHIP_DEFAULT_BIND_POSE ? bind_pose_of(clip) : params.default_values;
					const float* source = table + (mode != ACLHIP_DEFAULT_CONSTANT ? size_t(track_index) * 12 : 0) + kind * 4;
					value = f32x4{ source[0], source[1], source[2], kind == 0 ? source[3] : 0.0f };
				}
			}

			if (!kCompactOutput)
			{
				if (store)
					store_streaming(&pose[r * k_wave_size], value);
			}
			else
			{
				// aclhip_output_desc: sub-track kinds the writer skips (track_writer::skip_all_*, core/track_writer.h:181-183) are not
				// stored; the compact layouts drop the padding lanes (QVV40: rotation xyzw | translation xyz | scale xyz, what the
				// reference's benchmark counts per bone, tools/acl_decompressor/sources/benchmark.cpp:146) or the scale altogether
				// (QV32: rotation xyzw | translation xyz 0). Lanes keep their image quad: the stores of a row still cover one contiguous
				// span of the pose, minus the dropped pieces.
				const uint32_t track_index = (lane_quad + r * k_wave_size) / 3u;
				store = store && ((params.skip_mask >> kind) & 1u) == 0;
				if (skip_tracks != nullptr && store)
					store = ((skip_tracks[track_index] >> kind) & 1u) == 0;
				if (params.layout == ACLHIP_LAYOUT_QVV48)
				{
					if (store)
						store_streaming(&pose[r * k_wave_size], value);
				}
				else if (params.layout == ACLHIP_LAYOUT_QV32)
				{
					if (store && kind != 2)
						store_streaming(pose_bytes + size_t(track_index * 2u + kind) * 16, value);
				}
				else
				{
					uint8_t* address = pose_bytes + size_t(track_index) * 40 + (kind == 0 ? 0u : (kind == 1 ? 16u : 28u));
					if (store && kind == 0)
						store_streaming(address, value);
					if (store && kind != 0)
					{
						const float xyz[3] = { value.x, value.y, value.z };
						store_streaming_floats<3>(reinterpret_cast<float*>(address), xyz);
					}
				}
			}
			if (kAnySettings || kCompactOutput)
				kind = kind == 2 ? 0u : kind + 1u;
		}
		ACLHIP_WAVE0_STAMP(3);
	}

	// ---- compact output layouts, the fast path -----------------------------------------------------------------------------------------
	// aclhip_output_desc layouts QVV40 / QV32 with the reference defaults and nothing else skipped: the window's LDS image is built in
	// the OUTPUT layout from the start, so the way out is the same plain copy as for QVV48 -- 1 KiB of contiguous HBM per store
	// instruction -- with a third (QV32) or a sixth (QVV40) fewer bytes:
	//   QV32   rotation xyzw | translation xyz 0: the base pose DMA gathers quads 3t, 3t + 1 of the clip's resolved pose (global_load_lds
	//          takes a per lane source address), scale sub-tracks are decoded by nobody;
	//   QVV40  10 packed floats per track: the DMA source is a second resolved pose image registration lays out in that form behind the
	//          first (40 bytes per track); a decoded sub-track is written at float 10 t + {0, 4, 7} (8 byte aligned pieces).
	// Every other combination (skipped kinds, the other default modes, per track rounding) takes the generic compact kernels below.
	template<uint32_t kLayout>
	struct compact_image_writer
	{
		float* image;
		uint32_t first_track;
		uint32_t window_tracks;		// (see qvv48_image_writer)
		__device__ __forceinline__ void operator()(const clip_range_entry& entry, float4 value) const
		{
			typedef float f32x2 __attribute__((ext_vector_type(2)));
			const uint32_t kind = entry.quad_index - entry.track_index * 3u;
			const uint32_t track = entry.track_index - first_track;
			if (track >= window_tracks)
				return;
			if (kLayout == ACLHIP_LAYOUT_QV32)
			{
				if (kind != 2)
					reinterpret_cast<f32x4*>(image)[track * 2u + kind] = f32x4{ value.x, value.y, value.z, value.w };
				return;
			}
			float* record = image + track * 10u;		// 40 bytes per track: 8 byte aligned
			if (kind == 0)
			{
				*reinterpret_cast<f32x2*>(record) = f32x2{ value.x, value.y };
				*reinterpret_cast<f32x2*>(record + 2) = f32x2{ value.z, value.w };
			}
			else if (kind == 1)
			{
				*reinterpret_cast<f32x2*>(record + 4) = f32x2{ value.x, value.y };
				record[6] = value.z;
			}
			else
			{
				record[7] = value.x;
				*reinterpret_cast<f32x2*>(record + 8) = f32x2{ value.y, value.z };
			}
		}
	};

	template<uint32_t kLayout>
	__device__ __forceinline__ void decompress_tracks_compact_window(const device_clip* __restrict__ clips, uint32_t num_clips,
		const uint32_t* __restrict__ clip_ids, const float* __restrict__ sample_times, uint32_t num_instances, uint32_t windows_per_instance,
		const decode_params& params, uint8_t* __restrict__ poses, uint64_t pose_stride_bytes, uint32_t lds_quads_per_wave,
		unsigned long long* __restrict__ rejected_count)
	{
		static_assert(kLayout == ACLHIP_LAYOUT_QV32 || kLayout == ACLHIP_LAYOUT_QVV40, "compact layouts");
		extern __shared__ __attribute__((aligned(16))) uint8_t dynamic_lds[];
		ACLHIP_WAVE0_STAMP(0);
		constexpr uint32_t k_window_tracks = k_image_chunk_quads / 3u;
		constexpr uint32_t k_track_bytes = kLayout == ACLHIP_LAYOUT_QV32 ? 32u : 40u;

		const uint32_t lane = threadIdx.x & (k_wave_size - 1);
		const uint32_t wave_in_block = __builtin_amdgcn_readfirstlane(threadIdx.x / k_wave_size);
		const uint32_t work_item = blockIdx.x * k_waves_per_block + wave_in_block;
		uint32_t instance = work_item;
		uint32_t window = 0;
		if (windows_per_instance != 1)
		{
			instance = work_item / windows_per_instance;
			window = work_item - instance * windows_per_instance;
		}
		if (instance >= num_instances)
			return;

		instance_inputs inputs;
		load_instance_inputs(clip_ids, sample_times, params, instance, pose_stride_bytes, lds_quads_per_wave, true, inputs);
		const uint32_t clip_id = inputs.clip_id, caller_instance = inputs.caller_instance;
		const float sample_time = inputs.sample_time;
		const uint32_t rounding_policy = uniform_instance_rounding_policy_of(params, caller_instance, clips);		// (see decompress_tracks_window)
		const uint32_t looping_policy = uniform_instance_looping_policy_of(params, caller_instance, clips);
		const device_clip clip = load_clip(clips, clip_id < num_clips ? clip_id : 0);
		const uint32_t stored_tracks = stored_tracks_of(params, caller_instance, clip.num_tracks);		// (see decompress_tracks_window)
		if (clip_id >= num_clips || !is_transform_clip(clip.flags) || launch_refuses_clip(stored_tracks, windows_per_instance, lds_quads_per_wave, k_track_bytes, pose_stride_bytes))
		{
			if (lane == 0 && window == 0)
				atomicAdd(rejected_count, 1ull);
			return;
		}

		const uint32_t first_track = window * k_window_tracks;
		if (first_track >= stored_tracks)
			return;
		const uint32_t window_tracks = min(stored_tracks - first_track, k_window_tracks);
		const uint32_t window_bytes = window_tracks * k_track_bytes;
		const uint32_t window_pieces = (window_bytes + 15u) / 16u;		// QVV40 with an odd number of tracks ends on half a piece

		uint32_t first_ordinal = 0, end_ordinal = clip.num_animated;
		if (clip.num_tracks > k_window_tracks)
		{
			first_ordinal = as_constant(clip.image_chunks)[window];
			end_ordinal = as_constant(clip.image_chunks)[window + 1];
		}

		f32x4* image = reinterpret_cast<f32x4*>(dynamic_lds) + size_t(wave_in_block) * lds_quads_per_wave;

		// base pose -> LDS image, already in the output layout
		for (uint32_t base = 0; base < window_pieces; base += k_wave_size)
		{
			const uint32_t piece = base + lane;
			if (piece < window_pieces)
			{
				const ACLHIP_CONSTANT f32x4* source;
				if (kLayout == ACLHIP_LAYOUT_QV32)
					source = (const ACLHIP_CONSTANT f32x4*)clip.resolved_pose + (first_track + (piece >> 1)) * 3u + (piece & 1u);
				else
					source = (const ACLHIP_CONSTANT f32x4*)((const ACLHIP_CONSTANT uint8_t*)(clip.resolved_pose + size_t(clip.num_tracks) * 3u) + size_t(first_track) * 40u) + piece;
				__builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)source, (__attribute__((address_space(3))) void*)(image + base), 16, 0, 0);
			}
		}


		seek_state state;
		seek(clip, sample_time, rounding_policy, looping_policy, state);
#if defined(ACLHIP_EXP_PHASE_TIMES)
		asm volatile("" :: "s"(state.key_frame_bit_offsets[0]), "s"(state.key_frame_bit_offsets[1]));		// the seek's loads have arrived
		ACLHIP_WAVE0_STAMP(1);
#endif

		// the base pose must be in the image before decoded sub-tracks take their places in it (the QVV40 pieces of a decoded sub-track
		// and of its constant neighbours share 16 byte units: DMA first, then the decode's own writes)
		const compact_image_writer<kLayout> writer = { reinterpret_cast<float*>(image), first_track, window_tracks };
		decode_window_sub_tracks_into<false>(window_tables_of(clip), state, params, rounding_policy, params.normalization, first_ordinal, end_ordinal, lane, writer, nullptr, true);

		__builtin_amdgcn_s_waitcnt(0);
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
		__builtin_amdgcn_wave_barrier();
		__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

		ACLHIP_WAVE0_STAMP(2);
		constexpr uint32_t k_rows = (k_window_tracks * k_track_bytes / 16u + k_wave_size - 1) / k_wave_size;
		f32x4 staged[k_rows];
		#pragma unroll
		for (uint32_t r = 0; r < k_rows; ++r)
			staged[r] = image[min(r * k_wave_size + lane, window_pieces - 1)];

		const uint32_t row = params.instance_rows != nullptr ? as_constant(params.instance_rows)[instance] : instance;
		uint8_t* out = poses + uint64_t(row) * pose_stride_bytes + size_t(first_track) * k_track_bytes;
		const uint32_t whole_pieces = window_bytes / 16u;
		#pragma unroll
		for (uint32_t r = 0; r < k_rows; ++r)
		{
			const uint32_t piece = r * k_wave_size + lane;
			if (piece < whole_pieces)
				store_streaming(out + size_t(piece) * 16, staged[r]);
			else if (kLayout == ACLHIP_LAYOUT_QVV40 && piece < window_pieces)
			{
				const float half[2] = { staged[r].x, staged[r].y };
				store_streaming_floats<2>(reinterpret_cast<float*>(out + size_t(piece) * 16), half);
			}
		}
		ACLHIP_WAVE0_STAMP(3);
	}

	// (pose_kernel_args above mirrors this list as it lies in the kernarg segment: change both together -- tests/test_gpu_full_size.py's rig
	// batches decode garbage at once if they disagree)
	#define ACLHIP_POSE_KERNEL_ARGUMENTS const device_clip* __restrict__ clips, uint32_t num_clips, \
		const uint32_t* __restrict__ clip_ids, const float* __restrict__ sample_times, uint32_t num_instances, uint32_t windows_per_instance, \
		decode_params params, uint8_t* __restrict__ poses, uint64_t pose_stride_bytes, uint32_t lds_quads_per_wave, unsigned long long* __restrict__ rejected_count
	#define ACLHIP_POSE_KERNEL_FORWARD clips, num_clips, clip_ids, sample_times, num_instances, windows_per_instance, params, poses, pose_stride_bytes, lds_quads_per_wave, rejected_count

	// one wave per (instance, pose window): workgroup b holds work items 4 b .. 4 b + 3
	__device__ __forceinline__ uint32_t one_shot_work_item()
	{
		return blockIdx.x * k_waves_per_block + __builtin_amdgcn_readfirstlane(threadIdx.x / k_wave_size);
	}

	__global__ __launch_bounds__(k_block_size) __attribute__((amdgpu_waves_per_eu(8, 8))) void decompress_tracks_kernel(ACLHIP_POSE_KERNEL_ARGUMENTS)
	{
		decompress_tracks_window<false, false>(ACLHIP_POSE_KERNEL_FORWARD, one_shot_work_item());
	}

	// the same for poses of several windows (the 300-bone rig): one dword aligned 16 byte read of the bitstream per key
	// (unpack_animated_samples_wide, aclhip_device.h)
	__global__ __launch_bounds__(k_block_size) __attribute__((amdgpu_waves_per_eu(8, 8))) void decompress_tracks_wide_loads_kernel(ACLHIP_POSE_KERNEL_ARGUMENTS)
	{
		decompress_tracks_window<false, false, true>(ACLHIP_POSE_KERNEL_FORWARD, one_shot_work_item());
	}

	// Poses of several windows, since round 4: every wave takes params.items_per_wave work items IN TURN and keeps its LDS image from
	// one to the next. A wave's items are the same window of different instances, so while the clip stays the same (a crowd of one
	// rig: BASELINE.json configs[3]) the image already holds the clip's resolved pose window -- the animated quads are about to be
	// overwritten, the others are this clip's constants -- and the base pose copy, the largest single item through the CU's texture
	// unit (11 % of the launch, profiles/r03_experiments.md), is skipped from the second turn on. The next item's scalar seek also
	// hides the store acknowledgement of the item before. Until round 6 the loop needed 72 vector registers (one of them for 22 spilled
	// SGPRs) and ran at 7 waves per SIMD; with its arguments read from the kernarg segment per item it needs 64: 8 waves, what the
	// launch's LDS allows (32 waves per CU).
	//   kAdjacentItems = false: item k of workgroup b is work item 4 (k gridDim + b) + wave -- every turn sweeps the batch front to back
	//                           like the one-shot grid does (the host sizes the grid so that a wave keeps its window index);
	//   kAdjacentItems = true:  wave g takes window g % W of instances (g / W) K .. (g / W) K + K - 1: consecutive instances, which in a
	//                           list bucketed by clip are of one clip.
	template<bool kAdjacentItems, bool kWideKeyLoads = true, uint32_t kFastMath = 0>
	__device__ __forceinline__ void decompress_tracks_windows_in_turn(ACLHIP_POSE_KERNEL_ARGUMENTS)
	{
		const uint32_t wave_in_block = __builtin_amdgcn_readfirstlane(threadIdx.x / k_wave_size);
		const uint32_t items_per_wave = params.items_per_wave;
		uint32_t image_clip = 0xFFFFFFFFu;		// the clip whose base pose window the wave's LDS image holds
		const uint32_t wave = blockIdx.x * k_waves_per_block + wave_in_block;
		const uint32_t group = wave / windows_per_instance;
		const uint32_t window = wave - group * windows_per_instance;
		const ACLHIP_CONSTANT pose_kernel_args* kernel_args = (const ACLHIP_CONSTANT pose_kernel_args*)__builtin_amdgcn_kernarg_segment_ptr();
		for (uint32_t turn = 0; turn < items_per_wave; ++turn)
		{
			const uint32_t work_item = kAdjacentItems ? (group * items_per_wave + turn) * windows_per_instance + window
				: (turn * gridDim.x + blockIdx.x) * k_waves_per_block + wave_in_block;
			// (this item's arguments, read from the kernarg segment now: opaque per turn, so that they are not hoisted out of the loop
			// and kept -- spilled -- across the decodes; see pose_kernel_args)
#if defined(__HIP_DEVICE_COMPILE__)
			const ACLHIP_CONSTANT pose_kernel_args* args = kernel_args;
			asm volatile("" : "+s"(args));
			decompress_tracks_window<false, false, kWideKeyLoads, kFastMath>(args->clips, args->num_clips, args->clip_ids, args->sample_times, args->num_instances, args->windows_per_instance,
				args->params, nullptr, args->pose_stride_bytes, args->lds_quads_per_wave, nullptr, work_item, &image_clip, kernel_args);
#else
			decompress_tracks_window<false, false, kWideKeyLoads, kFastMath>(ACLHIP_POSE_KERNEL_FORWARD, work_item, &image_clip, kernel_args);		// (the host pass only has to compile)
#endif
			// (the window's LDS reads completed before its stores were issued: the next turn's DMA may overwrite the image)
		}
	}

	// Waves per SIMD. With the arguments read from the kernarg segment per item (pose_kernel_args) the exact kernel needs 64 VGPRs and
	// fits 8 waves (2 spilled SGPRs): 191.3 us on the rig against 194.6 at 7 (65 VGPRs, nothing spilled; before the kernarg reads: 72
	// VGPRs, 22 spilled SGPRs, 194.6 us as well -- the 30 spill instructions per window were not what the launch waits for) and 201 at 6
	// (profiles/r06_experiments.md 11). The ACLHIP_DECODE_FAST variant spills vector registers at 8 (194.4 us) and stays at 7 (191.5).
#if !defined(ACLHIP_IN_TURN_WAVES_PER_EU)
	#define ACLHIP_IN_TURN_WAVES_PER_EU 8
#endif
#if !defined(ACLHIP_IN_TURN_OTHER_WAVES_PER_EU)
	#define ACLHIP_IN_TURN_OTHER_WAVES_PER_EU 7
#endif
	__global__ __launch_bounds__(k_block_size) __attribute__((amdgpu_waves_per_eu(ACLHIP_IN_TURN_WAVES_PER_EU, ACLHIP_IN_TURN_WAVES_PER_EU))) void decompress_tracks_in_turn_kernel(ACLHIP_POSE_KERNEL_ARGUMENTS)
	{
		decompress_tracks_windows_in_turn<false>(ACLHIP_POSE_KERNEL_FORWARD);
	}

	__global__ __launch_bounds__(k_block_size) __attribute__((amdgpu_waves_per_eu(ACLHIP_IN_TURN_OTHER_WAVES_PER_EU, ACLHIP_IN_TURN_OTHER_WAVES_PER_EU))) void decompress_tracks_in_turn_adjacent_kernel(ACLHIP_POSE_KERNEL_ARGUMENTS)
	{
		decompress_tracks_windows_in_turn<true>(ACLHIP_POSE_KERNEL_FORWARD);
	}

	// ACLHIP_DECODE_FAST (opt in, aclhip_decompress_params::flags): the three plain kernels with the rotation arithmetic in the hardware's
	// 1 ulp forms. The one-window kernel sits on its write stream and gains little; the kernels of poses of several windows are bound by
	// VALU issue and get their instructions back.
	__global__ __launch_bounds__(k_block_size) __attribute__((amdgpu_waves_per_eu(8, 8))) void decompress_tracks_fast_kernel(ACLHIP_POSE_KERNEL_ARGUMENTS)
	{
		decompress_tracks_window<false, false, false, 2>(ACLHIP_POSE_KERNEL_FORWARD, one_shot_work_item());
	}

	__global__ __launch_bounds__(k_block_size) __attribute__((amdgpu_waves_per_eu(8, 8))) void decompress_tracks_wide_loads_fast_kernel(ACLHIP_POSE_KERNEL_ARGUMENTS)
	{
		decompress_tracks_window<false, false, true, 2>(ACLHIP_POSE_KERNEL_FORWARD, one_shot_work_item());
	}

	__global__ __launch_bounds__(k_block_size) __attribute__((amdgpu_waves_per_eu(ACLHIP_IN_TURN_OTHER_WAVES_PER_EU, ACLHIP_IN_TURN_OTHER_WAVES_PER_EU))) void decompress_tracks_in_turn_fast_kernel(ACLHIP_POSE_KERNEL_ARGUMENTS)
	{
		decompress_tracks_windows_in_turn<false, true, 2>(ACLHIP_POSE_KERNEL_FORWARD);
	}

	__global__ __launch_bounds__(k_block_size) __attribute__((amdgpu_waves_per_eu(ACLHIP_IN_TURN_OTHER_WAVES_PER_EU, ACLHIP_IN_TURN_OTHER_WAVES_PER_EU))) void decompress_tracks_in_turn_adjacent_fast_kernel(ACLHIP_POSE_KERNEL_ARGUMENTS)
	{
		decompress_tracks_windows_in_turn<true, true, 2>(ACLHIP_POSE_KERNEL_FORWARD);
	}

	// (One-window poses gain nothing from taking items in turn -- measured in round 5 with the byte-window key reads of the one-shot kernel,
	// 2 / 3 / 4 items per wave: headline 48.8 -> 53.7 / 56.2 / 57.3 us, 256 clips as drawn 62.0 -> 64.6 / 66.0 / 66.8, in locality order
	// 50.7 -> 54.1 / - / 55.7: at 8 waves per SIMD the one-shot grid already overlaps one wave's chain with the stores of seven others, and a
	// 100 bone pose pulls its base pose window out of the L2 either way. profiles/r05_experiments.md)

	// the common case with an aclhip_output_desc: compact layouts, skipped sub-track kinds
	__global__ __launch_bounds__(k_block_size) __attribute__((amdgpu_waves_per_eu(8, 8))) void decompress_tracks_compact_kernel(ACLHIP_POSE_KERNEL_ARGUMENTS)
	{
		decompress_tracks_window<false, true>(ACLHIP_POSE_KERNEL_FORWARD, one_shot_work_item());
	}

	__global__ __launch_bounds__(k_block_size) __attribute__((amdgpu_waves_per_eu(8, 8))) void decompress_tracks_qv32_kernel(ACLHIP_POSE_KERNEL_ARGUMENTS)
	{
		decompress_tracks_compact_window<ACLHIP_LAYOUT_QV32>(ACLHIP_POSE_KERNEL_FORWARD);
	}

	__global__ __launch_bounds__(k_block_size) __attribute__((amdgpu_waves_per_eu(8, 8))) void decompress_tracks_qvv40_kernel(ACLHIP_POSE_KERNEL_ARGUMENTS)
	{
		decompress_tracks_compact_window<ACLHIP_LAYOUT_QVV40>(ACLHIP_POSE_KERNEL_FORWARD);
	}

	// 8 waves per SIMD (64 VGPRs) matter more to this variant than the few instructions the allocator saves with 65
	__global__ __launch_bounds__(k_block_size) __attribute__((amdgpu_waves_per_eu(8, 8))) void decompress_tracks_any_settings_kernel(ACLHIP_POSE_KERNEL_ARGUMENTS)
	{
		decompress_tracks_window<true, false>(ACLHIP_POSE_KERNEL_FORWARD, one_shot_work_item());
	}

	__global__ __launch_bounds__(k_block_size) __attribute__((amdgpu_waves_per_eu(8, 8))) void decompress_tracks_any_settings_compact_kernel(ACLHIP_POSE_KERNEL_ARGUMENTS)
	{
		decompress_tracks_window<true, true>(ACLHIP_POSE_KERNEL_FORWARD, one_shot_work_item());
	}
