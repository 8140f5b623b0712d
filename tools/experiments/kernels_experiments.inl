// kernels_experiments.inl -- part of aclhip.hip, compiled only with -DACLHIP_EXPERIMENTS (tools/build_experiments.sh): the kernel variants
// round 3 built and MEASURED SLOWER than what ships (profiles/r03_experiments.md has their times and counters). Kept so that the
// measurements can be repeated; nothing here is reachable from a default build of libaclhip.so.
//   decompress_tracks_handoff_kernel / _last_arriver_kernel   decode waves hand finished LDS windows to a store wave / the last arriver (one-shot grids)
//   decompress_tracks_persistent_kernel                        a grid that fills the device once: looping decode waves + store waves around a pool of LDS images
//   decompress_tracks_in_turn_r3_kernel                           several work items per wave, the LDS image reused while the clip stays the same
//   decompress_tracks_staged_kernel                            keyframe bit runs staged through LDS, one base pose image per workgroup, merge on the way out

#if !defined(ACLHIP_DEFAULT_ITEMS_PER_WAVE)
	#define ACLHIP_DEFAULT_ITEMS_PER_WAVE 1
#endif
	constexpr uint32_t k_default_items_per_wave = ACLHIP_DEFAULT_ITEMS_PER_WAVE;

	// The same kernel with every wave taking SEVERAL work items in turn (params.items_per_wave; item k of workgroup b is work item
	// 4 (k gridDim + b) + wave: every turn sweeps the batch front to back like the one-shot grid does). Why: a wave cannot end before its
	// stores are acknowledged (s_endpgm waits for them) and holds its slot, registers and LDS image for that long -- ~1.7 of the
	// ~8.8 us a 300-bone window occupies a slot. A wave that moves on to its next item waits for nothing: the next item starts with
	// 2 us of SCALAR loads (the seek; their counter is lgkmcnt), and by the time it first waits for a vector load the stores of the
	// item before are long acknowledged. The hardware still balances the load: workgroups stay short (a few items) and plentiful.
	template<bool kAnySettings, bool kCompactOutput, bool kWideKeyLoads>
	__device__ __forceinline__ void decompress_tracks_windows_in_turn_r3(ACLHIP_POSE_KERNEL_ARGUMENTS)
	{
		const uint32_t wave_in_block = __builtin_amdgcn_readfirstlane(threadIdx.x / k_wave_size);
		const uint32_t items_per_wave = params.items_per_wave;
		// the clip whose base pose window the wave's LDS image holds (the host sizes the grid so that a wave keeps its window index from
		// turn to turn): the copy is skipped while the clip stays the same
		uint32_t image_clip = 0xFFFFFFFFu;
		for (uint32_t turn = 0; turn < items_per_wave; ++turn)
		{
			const uint32_t work_item = (turn * gridDim.x + blockIdx.x) * k_waves_per_block + wave_in_block;
			decompress_tracks_window<kAnySettings, kCompactOutput, kWideKeyLoads>(ACLHIP_POSE_KERNEL_FORWARD, work_item, &image_clip);
			// (the window's LDS reads completed before its stores were issued: the next turn's DMA may overwrite the image)
		}
	}

	__global__ __launch_bounds__(k_block_size) __attribute__((amdgpu_waves_per_eu(8, 8))) void decompress_tracks_in_turn_r3_kernel(ACLHIP_POSE_KERNEL_ARGUMENTS)
	{
		decompress_tracks_windows_in_turn_r3<false, false, false>(ACLHIP_POSE_KERNEL_FORWARD);
	}

	__global__ __launch_bounds__(k_block_size) __attribute__((amdgpu_waves_per_eu(8, 8))) void decompress_tracks_in_turn_wide_loads_kernel(ACLHIP_POSE_KERNEL_ARGUMENTS)
	{
		decompress_tracks_windows_in_turn_r3<false, false, true>(ACLHIP_POSE_KERNEL_FORWARD);
	}

	// round 4: the same with the register budget of 7 / 6 waves per SIMD (72 / 80 VGPRs): the loop around the 16 byte reads spills 25
	// registers at 64 (the launch is LDS limited to 32 waves per CU anyway)
	__global__ __launch_bounds__(k_block_size) __attribute__((amdgpu_waves_per_eu(7, 7))) void decompress_tracks_in_turn_wide_loads_7_kernel(ACLHIP_POSE_KERNEL_ARGUMENTS)
	{
		decompress_tracks_windows_in_turn_r3<false, false, true>(ACLHIP_POSE_KERNEL_FORWARD);
	}

	__global__ __launch_bounds__(k_block_size) __attribute__((amdgpu_waves_per_eu(6, 6))) void decompress_tracks_in_turn_wide_loads_6_kernel(ACLHIP_POSE_KERNEL_ARGUMENTS)
	{
		decompress_tracks_windows_in_turn_r3<false, false, true>(ACLHIP_POSE_KERNEL_FORWARD);
	}

	// ---- decode waves hand their windows to a store wave ---------------------------------------------------------------------------------
	// The common-case kernel above for poses of several windows (the 300-bone rig): a workgroup is kDecoders decode waves + ONE store
	// wave. A decode wave does everything decompress_tracks_window does up to the finished LDS image, publishes it (a descriptor + a
	// flag in LDS) and ENDS -- its wave slot and registers go to the next workgroup's decoders while its image waits; the store wave
	// takes finished images in whatever order they complete, LDS -> registers -> HBM, 1 KiB per store instruction as before.
	// Why: the HBM write path sustains 6.75 TB/s when <= 8 waves of a CU are inside their store phase and 5.5 TB/s with 32;
	// here at most (resident workgroups) waves of a CU ever issue stores, and no decode wave sits on its slot while its stores drain.
#if !defined(ACLHIP_HANDOFF_DEFAULT_DECODERS)
	#define ACLHIP_HANDOFF_DEFAULT_DECODERS 0
#endif
	constexpr uint32_t k_handoff_default_decoders = ACLHIP_HANDOFF_DEFAULT_DECODERS;

	struct alignas(16) handoff_descriptor
	{
		uint32_t ready;				// 0 until the image is complete (or the wave has nothing to store: window_quads = 0)
		uint32_t window_quads;
		uint32_t pose_offset_lo;	// byte offset of the window's first quad from `poses`
		uint32_t pose_offset_hi;
	};

	// kDedicatedStoreWave = false is the same hand-over WITHOUT the extra wave: every decoder takes a ticket when its image is done, all but
	// the last one end, and the LAST ARRIVER stores the workgroup's images -- no wave slot is spent on waiting.
	template<uint32_t kDecoders, bool kDedicatedStoreWave>
	__device__ __forceinline__ void decompress_tracks_handoff(const device_clip* __restrict__ clips, uint32_t num_clips,
		const uint32_t* __restrict__ clip_ids, const float* __restrict__ sample_times, uint32_t num_instances, uint32_t windows_per_instance,
		const decode_params& params, uint8_t* __restrict__ poses, uint64_t pose_stride_bytes, uint32_t lds_quads_per_wave,
		unsigned long long* __restrict__ rejected_count)
	{
		extern __shared__ __attribute__((aligned(16))) uint8_t dynamic_lds[];
		static_assert(kDecoders >= 1 && kDecoders <= 15, "decode waves of a workgroup: their ready flags are gathered by one ballot");

		const uint32_t lane = threadIdx.x & (k_wave_size - 1);
		const uint32_t wave_in_block = __builtin_amdgcn_readfirstlane(threadIdx.x / k_wave_size);
		handoff_descriptor* descriptors = reinterpret_cast<handoff_descriptor*>(dynamic_lds);		// [kDecoders]
		uint32_t* arrivals = reinterpret_cast<uint32_t*>(dynamic_lds + 240);
		f32x4* images = reinterpret_cast<f32x4*>(dynamic_lds + 256);

		// LDS is not cleared between workgroups: every decoder lowers its own flag, and nobody looks at a flag before this barrier
		if (wave_in_block < kDecoders && lane == 0)
			descriptors[wave_in_block].ready = 0;
		if (!kDedicatedStoreWave && threadIdx.x == 0)
			*arrivals = 0;
		__syncthreads();

		constexpr uint32_t k_rows = (k_image_chunk_quads + k_wave_size - 1) / k_wave_size;
		const auto store_image = [&](uint32_t decoder)
		{
			const uint32_t window_quads = __builtin_amdgcn_readfirstlane(descriptors[decoder].window_quads);
			if (window_quads == 0)
				return;
			const uint64_t pose_offset = uint64_t(__builtin_amdgcn_readfirstlane(descriptors[decoder].pose_offset_lo))
				| (uint64_t(__builtin_amdgcn_readfirstlane(descriptors[decoder].pose_offset_hi)) << 32);
			const f32x4* image = images + size_t(decoder) * lds_quads_per_wave;
			const uint32_t full_rows = window_quads / k_wave_size;
			f32x4 staged[k_rows];
			#pragma unroll
			for (uint32_t r = 0; r < k_rows; ++r)
				staged[r] = image[min(r * k_wave_size + lane, lds_quads_per_wave - 1)];
			f32x4* pose = reinterpret_cast<f32x4*>(poses + pose_offset) + lane;
			#pragma unroll
			for (uint32_t r = 0; r < k_rows; ++r)
				if (r < full_rows || (r == full_rows && r * k_wave_size + lane < window_quads))
					store_streaming(&pose[r * k_wave_size], staged[r]);
		};

		if (kDedicatedStoreWave && wave_in_block == kDecoders)
		{
			// ---- the store wave ----
			uint32_t pending = (1u << kDecoders) - 1u;
			while (pending != 0)
			{
				// lane w looks at decoder w's flag
				uint32_t flag = 0;
				if (lane < kDecoders)
					flag = __hip_atomic_load(&descriptors[lane].ready, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
				uint32_t ready = uint32_t(__ballot(flag != 0)) & pending;
				if (ready == 0)
				{
					__builtin_amdgcn_s_sleep(4);
					continue;
				}
				__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
				while (ready != 0)
				{
					const uint32_t decoder = uint32_t(__builtin_ctz(ready));
					ready &= ready - 1u;
					pending &= ~(1u << decoder);
					store_image(decoder);
				}
			}
			return;
		}

		// ---- a decode wave ----
		handoff_descriptor* descriptor = descriptors + wave_in_block;
		const auto publish = [&](uint32_t window_quads, uint64_t pose_offset)
		{
			// DMA and this wave's LDS writes have landed (s_waitcnt 0) before the flag goes up
			__builtin_amdgcn_s_waitcnt(0);
			__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
			uint32_t ticket = 0;
			if (lane == 0)
			{
				descriptor->window_quads = window_quads;
				descriptor->pose_offset_lo = uint32_t(pose_offset);
				descriptor->pose_offset_hi = uint32_t(pose_offset >> 32);
				__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
				if (kDedicatedStoreWave)
					__hip_atomic_store(&descriptor->ready, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
				else
					ticket = __hip_atomic_fetch_add(arrivals, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_WORKGROUP);
			}
			if (!kDedicatedStoreWave)
			{
				// the last arriver stores every image of the workgroup; everybody else is done
				ticket = __builtin_amdgcn_readfirstlane(ticket);
				if (ticket != kDecoders - 1u)
					return;
				__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
				for (uint32_t decoder = 0; decoder < kDecoders; ++decoder)
					store_image(decoder);
			}
		};

		const uint32_t work_item = blockIdx.x * kDecoders + wave_in_block;
		uint32_t instance = work_item;
		uint32_t window = 0;
		if (windows_per_instance != 1)
		{
			instance = work_item / windows_per_instance;
			window = work_item - instance * windows_per_instance;
		}
		if (instance >= num_instances)
		{
			publish(0, 0);
			return;
		}

		const uint32_t clip_id = as_constant(clip_ids)[instance];
		const float sample_time = as_constant(sample_times)[instance];
		const device_clip clip = load_clip(clips, clip_id < num_clips ? clip_id : 0);
		if (clip_id >= num_clips || !is_transform_clip(clip.flags))
		{
			if (lane == 0 && window == 0)
				atomicAdd(rejected_count, 1ull);
			publish(0, 0);
			return;
		}

		const uint32_t num_quads = clip.num_tracks * 3u;
		const uint32_t first_quad = window * k_image_chunk_quads;
		if (first_quad >= num_quads)
		{
			publish(0, 0);
			return;
		}
		const uint32_t window_quads = min(num_quads - first_quad, k_image_chunk_quads);

		uint32_t first_ordinal = 0, end_ordinal = clip.num_animated;
		if (num_quads > k_image_chunk_quads)
		{
			first_ordinal = as_constant(clip.image_chunks)[window];
			end_ordinal = as_constant(clip.image_chunks)[window + 1];
		}

		f32x4* image = images + size_t(wave_in_block) * lds_quads_per_wave;
		{
			const ACLHIP_CONSTANT f32x4* source = (const ACLHIP_CONSTANT f32x4*)clip.resolved_pose + first_quad;
			for (uint32_t base = 0; base < window_quads; base += k_wave_size)
			{
				if (base + lane < window_quads)
					__builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(source + base + lane),
						(__attribute__((address_space(3))) void*)(image + base), 16, 0, 0);
			}
		}

		const uint32_t rounding_policy = params.instance_rounding_policies != nullptr
			? __builtin_amdgcn_readfirstlane(uint32_t(params.instance_rounding_policies[instance]))
			: uint32_t(params.rounding_policy);

		seek_state state;
		seek(clip, sample_time, rounding_policy, params.looping_policy, state);

		decode_window_sub_tracks<false>(window_tables_of(clip), state, params, rounding_policy, params.normalization, first_ordinal, end_ordinal, first_quad, window_quads, lane, image);

		const uint32_t row = params.instance_rows != nullptr ? as_constant(params.instance_rows)[instance] : instance;
		publish(window_quads, uint64_t(row) * pose_stride_bytes + uint64_t(first_quad) * 16u);
	}

	template<uint32_t kDecoders>
	__global__ __launch_bounds__((kDecoders + 1) * k_wave_size) __attribute__((amdgpu_waves_per_eu(8, 8))) void decompress_tracks_handoff_kernel(ACLHIP_POSE_KERNEL_ARGUMENTS)
	{
		decompress_tracks_handoff<kDecoders, true>(ACLHIP_POSE_KERNEL_FORWARD);
	}

	template<uint32_t kDecoders>
	__global__ __launch_bounds__(kDecoders * k_wave_size) __attribute__((amdgpu_waves_per_eu(8, 8))) void decompress_tracks_last_arriver_kernel(ACLHIP_POSE_KERNEL_ARGUMENTS)
	{
		decompress_tracks_handoff<kDecoders, false>(ACLHIP_POSE_KERNEL_FORWARD);
	}

	// ---- persistent decode waves + store waves --------------------------------------------------------------------------------------------
	// A wave cannot end before its stores are acknowledged (s_endpgm waits for them), and a wave that keeps going cannot tell its loads
	// from its stores (one counter, vmcnt): either way a wave that stores its own window sits on its slot, its registers and its LDS
	// image for the 2 - 3 us the write takes to be acknowledged -- a third of its life on the 300-bone rig (tools/write_probe5.hip:
	// a wave's fixed cost around five 1 KiB stores is ~3.4 us). Here nothing that decodes ever stores:
	//   * the grid fills the device ONCE; a workgroup is kDecoders decode waves + kStorers store waves around a pool of kSlots LDS images;
	//   * a decode wave loops over its work items: take a free image, DMA the base pose window into it, seek, decode, publish the
	//     image (descriptor + a bit in ready_mask), take the next free image. Its vmcnt only ever counts loads;
	//   * a store wave loops: claim a ready image, LDS -> registers, give the image back (free_mask), 1 KiB stores. It never waits for an
	//     acknowledgement -- up to 63 stores stay in flight per wave -- until it ends.
	// Work item i of iteration k belongs to decoder (i mod total decoders): every iteration sweeps one contiguous range of the batch,
	// like the hardware's dispatch order does for the one-shot kernels.
#if !defined(ACLHIP_PERSISTENT_DEFAULT_SHAPE)
	#define ACLHIP_PERSISTENT_DEFAULT_SHAPE 0
#endif
	constexpr uint32_t k_persistent_default_shape = ACLHIP_PERSISTENT_DEFAULT_SHAPE;

	struct alignas(16) slot_descriptor
	{
		uint32_t window_quads;
		uint32_t pose_offset_lo;	// byte offset of the window's first quad from `poses`
		uint32_t pose_offset_hi;
		uint32_t reserved;
	};

	struct alignas(16) persistent_control		// the first 512 bytes of the workgroup's LDS
	{
		uint32_t ready_mask;		// images that are complete and wait for a store wave
		uint32_t free_mask;			// images nobody owns
		uint32_t decoders_done;		// decode waves that ran out of work
		uint32_t reserved;
		slot_descriptor slots[31];
	};
	static_assert(sizeof(persistent_control) == 512, "layout");

	template<uint32_t kDecoders, uint32_t kStorers, uint32_t kSlots>
	__device__ __forceinline__ void decompress_tracks_persistent(const device_clip* __restrict__ clips, uint32_t num_clips,
		const uint32_t* __restrict__ clip_ids, const float* __restrict__ sample_times, uint32_t num_instances, uint32_t windows_per_instance,
		const decode_params& params, uint8_t* __restrict__ poses, uint64_t pose_stride_bytes, uint32_t lds_quads_per_wave,
		unsigned long long* __restrict__ rejected_count)
	{
		extern __shared__ __attribute__((aligned(16))) uint8_t dynamic_lds[];
		static_assert(kSlots > kDecoders && kSlots <= 31, "every decoder owns an image while it decodes; the rest wait to be stored");

		const uint32_t lane = threadIdx.x & (k_wave_size - 1);
		const uint32_t wave_in_block = __builtin_amdgcn_readfirstlane(threadIdx.x / k_wave_size);
		persistent_control* control = reinterpret_cast<persistent_control*>(dynamic_lds);
		f32x4* images = reinterpret_cast<f32x4*>(dynamic_lds + sizeof(persistent_control));

		if (threadIdx.x == 0)
		{
			control->ready_mask = 0;
			control->free_mask = ((1u << kSlots) - 1u) & ~((1u << kDecoders) - 1u);		// decoder d starts on image d
			control->decoders_done = 0;
		}
		__syncthreads();

		// one lane talks to the control words, everybody gets the answer
		const auto claim_lowest_bit = [&](uint32_t* mask) -> uint32_t		// returns the claimed bit, or 0 when the mask was empty / somebody was faster
		{
			uint32_t claimed = 0;
			if (lane == 0)
			{
				const uint32_t seen = __hip_atomic_load(mask, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
				const uint32_t bit = seen & (0u - seen);
				if (bit != 0 && (__hip_atomic_fetch_and(mask, ~bit, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) & bit) != 0)
					claimed = bit;
			}
			return __builtin_amdgcn_readfirstlane(claimed);
		};

		if (wave_in_block >= kDecoders)
		{
			// ---- a store wave ----
			constexpr uint32_t k_rows = (k_image_chunk_quads + k_wave_size - 1) / k_wave_size;
			for (;;)
			{
				const uint32_t bit = claim_lowest_bit(&control->ready_mask);
				if (bit == 0)
				{
					// nothing ready: done when every decoder has left and nothing was published in between
					uint32_t finished = 0;
					if (lane == 0)
						finished = __hip_atomic_load(&control->decoders_done, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) == kDecoders
							&& __hip_atomic_load(&control->ready_mask, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == 0 ? 1u : 0u;
					if (__builtin_amdgcn_readfirstlane(finished) != 0)
						break;
					__builtin_amdgcn_s_sleep(2);
					continue;
				}
				const uint32_t slot = uint32_t(__builtin_ctz(bit));
				const uint32_t window_quads = __builtin_amdgcn_readfirstlane(control->slots[slot].window_quads);
				const uint64_t pose_offset = uint64_t(__builtin_amdgcn_readfirstlane(control->slots[slot].pose_offset_lo))
					| (uint64_t(__builtin_amdgcn_readfirstlane(control->slots[slot].pose_offset_hi)) << 32);
				const f32x4* image = images + size_t(slot) * lds_quads_per_wave;
				const uint32_t full_rows = window_quads / k_wave_size;
				f32x4 staged[k_rows];
				#pragma unroll
				for (uint32_t r = 0; r < k_rows; ++r)
					staged[r] = image[min(r * k_wave_size + lane, lds_quads_per_wave - 1)];
				// the image is in registers: a decoder may have it
				__builtin_amdgcn_s_waitcnt(0xC07F);		// lgkmcnt(0) only: never wait for the stores in flight (vmcnt)
				asm volatile("" :: "v"(staged[0]), "v"(staged[k_rows - 1]) : "memory");
				if (lane == 0)
					__hip_atomic_fetch_or(&control->free_mask, bit, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
				f32x4* pose = reinterpret_cast<f32x4*>(poses + pose_offset) + lane;
				#pragma unroll
				for (uint32_t r = 0; r < k_rows; ++r)
					if (r < full_rows || (r == full_rows && r * k_wave_size + lane < window_quads))
						store_streaming(&pose[r * k_wave_size], staged[r]);
			}
			return;
		}

		// ---- a decode wave ----
		const uint32_t total_decoders = gridDim.x * kDecoders;
		const uint64_t num_items = uint64_t(num_instances) * windows_per_instance;
		uint32_t slot = wave_in_block;
#if defined(ACLHIP_EXP_PHASE_TIMES)
		unsigned long long stamp_seek = 0, stamp_decode = 0, stamp_acquire = 0, stamp_items = 0;
#define ACLHIP_PERSISTENT_STAMP(total) do { const unsigned long long now = wall_clock64(); total += now - stamp_last; stamp_last = now; } while (0)
		unsigned long long stamp_last = wall_clock64();
		const unsigned long long stamp_first = stamp_last;
#else
#define ACLHIP_PERSISTENT_STAMP(total) do { } while (0)
#endif
		for (uint64_t item = uint64_t(blockIdx.x) * kDecoders + wave_in_block; item < num_items; item += total_decoders)
		{
			uint32_t instance = uint32_t(item);
			uint32_t window = 0;
			if (windows_per_instance != 1)
			{
				instance = uint32_t(item / windows_per_instance);
				window = uint32_t(item - uint64_t(instance) * windows_per_instance);
			}

			const uint32_t clip_id = as_constant(clip_ids)[instance];
			const float sample_time = as_constant(sample_times)[instance];
			const device_clip clip = load_clip(clips, clip_id < num_clips ? clip_id : 0);
			if (clip_id >= num_clips || !is_transform_clip(clip.flags))
			{
				if (lane == 0 && window == 0)
					atomicAdd(rejected_count, 1ull);
				continue;
			}

			const uint32_t num_quads = clip.num_tracks * 3u;
			const uint32_t first_quad = window * k_image_chunk_quads;
			if (first_quad >= num_quads)
				continue;
			const uint32_t window_quads = min(num_quads - first_quad, k_image_chunk_quads);

			uint32_t first_ordinal = 0, end_ordinal = clip.num_animated;
			if (num_quads > k_image_chunk_quads)
			{
				first_ordinal = as_constant(clip.image_chunks)[window];
				end_ordinal = as_constant(clip.image_chunks)[window + 1];
			}

			f32x4* image = images + size_t(slot) * lds_quads_per_wave;
			{
				const ACLHIP_CONSTANT f32x4* source = (const ACLHIP_CONSTANT f32x4*)clip.resolved_pose + first_quad;
				for (uint32_t base = 0; base < window_quads; base += k_wave_size)
				{
					if (base + lane < window_quads)
						__builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(source + base + lane),
							(__attribute__((address_space(3))) void*)(image + base), 16, 0, 0);
				}
			}

			const uint32_t rounding_policy = params.instance_rounding_policies != nullptr
				? __builtin_amdgcn_readfirstlane(uint32_t(params.instance_rounding_policies[instance]))
				: uint32_t(params.rounding_policy);

			seek_state state;
			seek(clip, sample_time, rounding_policy, params.looping_policy, state);
#if defined(ACLHIP_EXP_PHASE_TIMES)
			asm volatile("" :: "s"(state.key_frame_bit_offsets[0]), "s"(state.key_frame_bit_offsets[1]));		// the seek's loads have arrived
			ACLHIP_PERSISTENT_STAMP(stamp_seek);
			stamp_items++;
#endif

			decode_window_sub_tracks<false>(window_tables_of(clip), state, params, rounding_policy, params.normalization, first_ordinal, end_ordinal, first_quad, window_quads, lane, image);

			const uint32_t row = params.instance_rows != nullptr ? as_constant(params.instance_rows)[instance] : instance;
			const uint64_t pose_offset = uint64_t(row) * pose_stride_bytes + uint64_t(first_quad) * 16u;

			// the DMA and this wave's LDS writes have landed before the image is published
			__builtin_amdgcn_s_waitcnt(0);
			ACLHIP_PERSISTENT_STAMP(stamp_decode);
			__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
			if (lane == 0)
			{
				control->slots[slot].window_quads = window_quads;
				control->slots[slot].pose_offset_lo = uint32_t(pose_offset);
				control->slots[slot].pose_offset_hi = uint32_t(pose_offset >> 32);
				__hip_atomic_fetch_or(&control->ready_mask, 1u << slot, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
			}

			// the next image (one is always on its way back: kSlots > kDecoders)
			uint32_t bit;
			while ((bit = claim_lowest_bit(&control->free_mask)) == 0)
				__builtin_amdgcn_s_sleep(1);
			slot = uint32_t(__builtin_ctz(bit));
			ACLHIP_PERSISTENT_STAMP(stamp_acquire);
		}
		if (lane == 0)
			__hip_atomic_fetch_add(&control->decoders_done, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
#if defined(ACLHIP_EXP_PHASE_TIMES)
		if (lane == 0)
		{
			const uint32_t decoder = blockIdx.x * kDecoders + wave_in_block;
			if (decoder < 16384 / 2)
			{
				phase_times[decoder * 8 + 0] = stamp_seek; phase_times[decoder * 8 + 1] = stamp_decode; phase_times[decoder * 8 + 2] = stamp_acquire;
				phase_times[decoder * 8 + 3] = stamp_items; phase_times[decoder * 8 + 4] = stamp_first; phase_times[decoder * 8 + 5] = wall_clock64();
			}
		}
#endif
	}

	template<uint32_t kDecoders, uint32_t kStorers, uint32_t kSlots>
	__global__ __launch_bounds__((kDecoders + kStorers) * k_wave_size) __attribute__((amdgpu_waves_per_eu(8, 8))) void decompress_tracks_persistent_kernel(ACLHIP_POSE_KERNEL_ARGUMENTS)
	{
		decompress_tracks_persistent<kDecoders, kStorers, kSlots>(ACLHIP_POSE_KERNEL_FORWARD);
	}

	// ---- the staged kernel: keyframe bits staged through LDS, one base pose image per workgroup -------------------------------------------
	// The common case (track_writer defaults, no per track rounding, QVV48 poses) once more, built around what round 3's measurements
	// say the one-wave-per-window kernel above spends its time on -- the texture unit, busy 83 - 93 % of the 300-bone rig's launch:
	//   * a lane's four unaligned, scattered reads of the bitstream per pass (8 + 4 bytes per keyframe) cost the unit far more than
	//     their bytes: the same four loads from consecutive aligned addresses take 8.5 % off the launch;
	//   * every wave copies its window of the clip's base pose into LDS: another 11 %.
	// Here
	//   1. a workgroup is kWaves CONSECUTIVE INSTANCES of ONE pose window (workgroup b: window b mod W of instances kWaves (b / W) ..),
	//      and its first wave copies the clip's base pose window into LDS once, for all of them (instances of another clip read the base
	//      pose from memory when they store: the fallback of mixed batches);
	//   2. the runs of keyframe bits the window's animated sub-tracks cover (window_span_entry: one run per sub-track kind) travel
	//      global -> LDS in 16 byte pieces, lanes <-> consecutive pieces, and lanes pick their fields out of LDS (two aligned dwords and a
	//      64 bit shift per component);
	//   3. a wave's own LDS holds its window's DECODED sub-tracks only (and the staged bits): a third of a window image;
	//   4. on the way out lanes <-> consecutive pose quads take the base pose quad from the shared image and, where its W lane carries an
	//      animated sub-track's tag (the clip's tagged base pose: aclhip_device.h), the decoded value instead; default sub-tracks get their
	//      W resolved (the track_writer defaults). 1 KiB of contiguous HBM per store instruction, as before.
	// Same arithmetic, same bits out as decompress_tracks_kernel.
#if !defined(ACLHIP_DEFAULT_STAGED_WAVES)
	#define ACLHIP_DEFAULT_STAGED_WAVES 0
#endif
	constexpr uint32_t k_default_staged_waves = ACLHIP_DEFAULT_STAGED_WAVES;

	// The staged kernel's passes: lanes <-> the window's animated sub-tracks, 64 per pass. Compiled twice: most sample times fall between
	// two keyframes of ONE segment, whose table row then serves both keys (written as one loop with `entry1 = single ? entry0 : load` the
	// copy makes the wave wait for the row before it requests anything else).
	struct staged_biases { uint32_t rotation0, translation0, scale0, rotation1, translation1, scale1; };

	template<bool kSingleSegment>
	__device__ __forceinline__ void staged_decode_passes(const plan_entry* plan_row0, const plan_entry* plan_row1, const clip_range_entry* clip_ranges,
		const seek_state& state, uint32_t normalization, uint32_t first_ordinal, uint32_t end_ordinal, uint32_t lane, bool shares_base,
		lds_bytes key_bytes0, lds_bytes key_bytes1, staged_biases bias, f32x4* decoded)
	{
		uint32_t ordinal = min(first_ordinal + lane, end_ordinal - 1);
		plan_entry entry0 = load_entry(plan_row0, ordinal);
		plan_entry entry1_loaded;
		if constexpr (!kSingleSegment)
			entry1_loaded = load_entry(plan_row1, ordinal);
		clip_range_entry clip_range = load_entry(clip_ranges, ordinal);

		// the staged bits (and the first wave's base pose copy, requested before them) have landed ...
		__builtin_amdgcn_s_waitcnt(0);
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
		__builtin_amdgcn_wave_barrier();
		__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
		// ... for everybody who is going to read the shared copy (waves that left above are no longer counted by the barrier; all of a
		// sharing workgroup's waves get here, or none)
		if (shares_base)
			__syncthreads();

		for (uint32_t base = first_ordinal; base < end_ordinal; base += k_wave_size)
		{
			const bool valid = base + lane < end_ordinal;
			const plan_entry plan0 = entry0;
			const plan_entry& plan1 = kSingleSegment ? plan0 : entry1_loaded;
			const clip_range_entry current_range = clip_range;
			const uint32_t current_ordinal = ordinal;

			const uint32_t kind = current_range.quad_index - current_range.track_index * 3u;
			const bool is_rotation = kind == 0;
			const uint32_t num_bits0 = plan0.bit_offset_and_width >> 24, num_bits1 = plan1.bit_offset_and_width >> 24;
			// (a sub-track that is constant in its segment reads nothing: any position inside the buffer will do)
			const uint32_t position0 = num_bits0 == 0 ? 0u : (kind == 0 ? bias.rotation0 : (kind == 1 ? bias.translation0 : bias.scale0)) + (plan0.bit_offset_and_width & 0x00FFFFFFu);
			const uint32_t position1 = num_bits1 == 0 ? 0u : (kind == 0 ? bias.rotation1 : (kind == 1 ? bias.translation1 : bias.scale1)) + (plan1.bit_offset_and_width & 0x00FFFFFFu);

			const bool has_raw = __any(int(num_bits0 == 32u || num_bits1 == 32u)) != 0;
			float v0[3], v1[3];
			if (!has_raw)
				unpack_staged_samples<false>(key_bytes0, key_bytes1, position0, position1, plan0, plan1, current_range, is_rotation, v0, v1);
			else
				unpack_staged_samples<true>(key_bytes0, key_bytes1, position0, position1, plan0, plan1, current_range, is_rotation, v0, v1);
			const float4 value = interpolate_animated_samples<false>(state, v0, v1, is_rotation, k_round_none, state.interpolation_alpha, normalization, false, false);

			if (valid)
				decoded[current_ordinal - first_ordinal] = f32x4{ value.x, value.y, value.z, value.w };

			if (base + k_wave_size < end_ordinal)
			{
				ordinal = min(base + k_wave_size + lane, end_ordinal - 1);
				entry0 = load_entry(plan_row0, ordinal);
				if constexpr (!kSingleSegment)
					entry1_loaded = load_entry(plan_row1, ordinal);
				clip_range = load_entry(clip_ranges, ordinal);
			}
		}

		// this wave's decoded sub-tracks are in LDS for all its lanes
		__builtin_amdgcn_s_waitcnt(0);
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
		__builtin_amdgcn_wave_barrier();
		__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
	}

	#define ACLHIP_STAGED_KERNEL_ARGUMENTS ACLHIP_POSE_KERNEL_ARGUMENTS, uint32_t decoded_quads_per_wave, uint32_t key_bytes_per_wave

	template<uint32_t kWaves>
	__global__ __launch_bounds__(kWaves * k_wave_size) __attribute__((amdgpu_waves_per_eu(8, 8))) void decompress_tracks_staged_kernel(ACLHIP_STAGED_KERNEL_ARGUMENTS)
	{
		extern __shared__ __attribute__((aligned(16))) uint8_t dynamic_lds[];
		static_assert(kWaves == 4 || kWaves == 8, "the workgroup's clip handles arrive in one scalar load");

		const uint32_t lane = threadIdx.x & (k_wave_size - 1);
		const uint32_t wave_in_block = __builtin_amdgcn_readfirstlane(threadIdx.x / k_wave_size);
		uint32_t group = blockIdx.x, window = 0;
		if (windows_per_instance != 1)
		{
			group = blockIdx.x / windows_per_instance;
			window = blockIdx.x - group * windows_per_instance;
		}
		const uint32_t leader_instance = group * kWaves;		// < num_instances by the grid's size
		const uint32_t instance = leader_instance + wave_in_block;

		f32x4* shared_image = reinterpret_cast<f32x4*>(dynamic_lds);
		uint8_t* wave_lds = dynamic_lds + size_t(lds_quads_per_wave) * 16 + size_t(wave_in_block) * (size_t(decoded_quads_per_wave) * 16 + size_t(key_bytes_per_wave) * 2);
		f32x4* decoded = reinterpret_cast<f32x4*>(wave_lds);
		lds_bytes key_bytes0 = (lds_bytes)(wave_lds + size_t(decoded_quads_per_wave) * 16);
		lds_bytes key_bytes1 = key_bytes0 + key_bytes_per_wave;

		// Do the workgroup's instances all play ONE clip? Then its first wave's copy of the base pose window serves them all (every wave
		// comes to the same answer from the same scalar load: the decision needs no communication). Otherwise -- mixed batches, the
		// batch's tail -- every wave reads its own clip's base pose from memory when it stores.
		bool shares_base = leader_instance + kWaves <= num_instances;
		uint32_t clip_id = 0xFFFFFFFFu;
		if (shares_base)
		{
			typedef uint32_t handles_type __attribute__((ext_vector_type(kWaves)));
			const handles_type handles = *(const ACLHIP_CONSTANT handles_type*)(as_constant(clip_ids) + leader_instance);		// 16 / 32 byte aligned: leader_instance is a multiple of kWaves
			#pragma unroll
			for (uint32_t w = 0; w < kWaves; ++w)
			{
				shares_base = shares_base && handles[w] == handles[0];
				clip_id = w == wave_in_block ? handles[w] : clip_id;
			}
		}
		else
		{
			if (instance >= num_instances)
				return;
			clip_id = as_constant(clip_ids)[instance];
		}

		const float sample_time = as_constant(sample_times)[instance];
		const device_clip clip = load_clip(clips, clip_id < num_clips ? clip_id : 0);
		if (clip_id >= num_clips || !is_transform_clip(clip.flags))
		{
			if (lane == 0 && window == 0)
				atomicAdd(rejected_count, 1ull);
			return;		// (with a shared base the whole workgroup leaves here: one clip)
		}

		const uint32_t num_quads = clip.num_tracks * 3u;
		const uint32_t first_quad = window * k_image_chunk_quads;
		if (first_quad >= num_quads)
			return;
		const uint32_t window_quads = min(num_quads - first_quad, k_image_chunk_quads);
		const uint32_t num_windows = num_pose_windows(clip.num_tracks);

		uint32_t first_ordinal = 0, end_ordinal = clip.num_animated;
		if (num_quads > k_image_chunk_quads)
		{
			first_ordinal = as_constant(clip.image_chunks)[window];
			end_ordinal = as_constant(clip.image_chunks)[window + 1];
		}

		// 1. the first wave's copy of the base pose window (tagged: animated and default sub-tracks carry their marker in the W lane)
		if (shares_base && wave_in_block == 0)
		{
			const ACLHIP_CONSTANT f32x4* source = (const ACLHIP_CONSTANT f32x4*)clip.base_pose + first_quad;
			for (uint32_t base = 0; base < window_quads; base += k_wave_size)
			{
				if (base + lane < window_quads)
					__builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(source + base + lane),
						(__attribute__((address_space(3))) void*)(shared_image + base), 16, 0, 0);
			}
		}

		const uint32_t rounding_policy = params.instance_rounding_policies != nullptr
			? __builtin_amdgcn_readfirstlane(uint32_t(params.instance_rounding_policies[instance]))
			: uint32_t(params.rounding_policy);
		const uint32_t normalization = params.normalization;

		seek_state state;
		seek(clip, sample_time, rounding_policy, params.looping_policy, state);

		const bool has_animated = first_ordinal < end_ordinal;
		const plan_entry* plan_row0 = clip.plan + size_t(state.segment_index[0]) * clip.num_animated;
		const plan_entry* plan_row1 = clip.plan + size_t(state.segment_index[1]) * clip.num_animated;
		// (six named scalars, not an array: indexed by a lane's kind an array ends up in scratch memory)
		uint32_t bias_rotation0 = 0, bias_translation0 = 0, bias_scale0 = 0, bias_rotation1 = 0, bias_translation1 = 0, bias_scale1 = 0;
		if (has_animated)
		{
			// 2. both keyframes' runs of bits, global -> LDS: lanes <-> consecutive 16 byte pieces of the three runs laid end to end, one
			// request per key and 64 pieces. bias[key][kind]: what a lane adds to its sub-track's bit offset inside the keyframe to get
			// its bit position inside the key's staging buffer
			const ACLHIP_CONSTANT window_span_entry* spans = (const ACLHIP_CONSTANT window_span_entry*)(as_constant(clip.image_chunks) + window_spans_word_offset(num_windows));
			typedef uint32_t u32x8 __attribute__((ext_vector_type(8)));
			const u32x8 raw_span0 = ((const ACLHIP_CONSTANT u32x8*)spans)[size_t(state.segment_index[0]) * num_windows + window];		// one s_load_dwordx8 each
			const u32x8 raw_span1 = ((const ACLHIP_CONSTANT u32x8*)spans)[size_t(state.segment_index[1]) * num_windows + window];
			#pragma unroll
			for (uint32_t key = 0; key < 2; ++key)
			{
				window_span_entry span;
				__builtin_memcpy(&span, key == 0 ? &raw_span0 : &raw_span1, sizeof(span));
				const uint8_t* data = state.animated_track_data[key];
				const uint32_t misalignment = uint32_t(reinterpret_cast<uintptr_t>(data)) & 15u;
				const ACLHIP_CONSTANT f32x4* aligned_data = (const ACLHIP_CONSTANT f32x4*)(data - misalignment);
				const uint32_t key_bit = misalignment * 8u + state.key_frame_bit_offsets[key];		// of the keyframe's first bit, from aligned_data
				lds_bytes key_lds = key == 0 ? key_bytes0 : key_bytes1;

				// piece of the bitstream a run starts with; pieces it takes (+ one of slack: lanes read 8 byte windows), 0 for an empty run
#define ACLHIP_FIRST_PIECE_OF(kind) __builtin_amdgcn_readfirstlane((key_bit + span.first_bit[kind]) >> 7)
#define ACLHIP_NUM_PIECES_OF(kind) __builtin_amdgcn_readfirstlane(span.end_bit[kind] == span.first_bit[kind] ? 0u : ((key_bit + span.end_bit[kind] - 1u) >> 7) - ((key_bit + span.first_bit[kind]) >> 7) + 2u)
				const uint32_t first_rotation = ACLHIP_FIRST_PIECE_OF(0), first_translation = ACLHIP_FIRST_PIECE_OF(1), first_scale = ACLHIP_FIRST_PIECE_OF(2);
				const uint32_t end_rotation = ACLHIP_NUM_PIECES_OF(0), end_translation = end_rotation + ACLHIP_NUM_PIECES_OF(1), staged_pieces = end_translation + ACLHIP_NUM_PIECES_OF(2);
#undef ACLHIP_FIRST_PIECE_OF
#undef ACLHIP_NUM_PIECES_OF
				const uint32_t bias_rotation = key_bit - (first_rotation << 7);
				const uint32_t bias_translation = end_rotation * 128u + key_bit - (first_translation << 7);
				const uint32_t bias_scale = end_translation * 128u + key_bit - (first_scale << 7);
				if (key == 0)
				{
					bias_rotation0 = bias_rotation; bias_translation0 = bias_translation; bias_scale0 = bias_scale;
				}
				else
				{
					bias_rotation1 = bias_rotation; bias_translation1 = bias_translation; bias_scale1 = bias_scale;
				}
				for (uint32_t base = 0; base < staged_pieces; base += k_wave_size)
				{
					const uint32_t piece = base + lane;
					const uint32_t source_piece = piece < end_rotation ? first_rotation + piece
						: (piece < end_translation ? first_translation + (piece - end_rotation) : first_scale + (piece - end_translation));
					if (piece < staged_pieces)
						__builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(aligned_data + source_piece),
							(__attribute__((address_space(3))) void*)(key_lds + size_t(base) * 16), 16, 0, 0);
				}
			}
		}

		// 3. lanes <-> the window's animated sub-tracks, 64 per pass
		if (!has_animated)
		{
			// nothing animated in this window: the base pose copy is all there is
			__builtin_amdgcn_s_waitcnt(0);
			if (shares_base)
				__syncthreads();
		}
		else
		{
			const staged_biases bias = { bias_rotation0, bias_translation0, bias_scale0, bias_rotation1, bias_translation1, bias_scale1 };
			if (state.uses_single_segment)
				staged_decode_passes<true>(plan_row0, plan_row1, clip.clip_ranges, state, normalization, first_ordinal, end_ordinal, lane, shares_base, key_bytes0, key_bytes1, bias, decoded);
			else
				staged_decode_passes<false>(plan_row0, plan_row1, clip.clip_ranges, state, normalization, first_ordinal, end_ordinal, lane, shares_base, key_bytes0, key_bytes1, bias, decoded);
		}

		// 4. lanes <-> consecutive pose quads: base pose quad, or the decoded sub-track its tag names
		constexpr uint32_t k_rows = (k_image_chunk_quads + k_wave_size - 1) / k_wave_size;
		const uint32_t full_rows = window_quads / k_wave_size;
		f32x4 staged[k_rows];
		if (shares_base)
		{
			#pragma unroll
			for (uint32_t r = 0; r < k_rows; ++r)
				staged[r] = shared_image[min(r * k_wave_size + lane, lds_quads_per_wave - 1)];
		}
		else
		{
			const ACLHIP_CONSTANT f32x4* source = (const ACLHIP_CONSTANT f32x4*)clip.base_pose + first_quad;
			#pragma unroll
			for (uint32_t r = 0; r < k_rows; ++r)
				staged[r] = source[min(r * k_wave_size + lane, window_quads - 1)];
		}

		// (every lane reads a decoded sub-track per row -- its own where its quad is animated, the window's first otherwise -- so that the
		// five LDS reads travel together instead of one per branch)
		f32x4 animated[k_rows];
		#pragma unroll
		for (uint32_t r = 0; r < k_rows; ++r)
		{
			const uint32_t marker = __float_as_uint(staged[r].w);
			const bool is_animated = (marker & (k_quad_special | k_quad_animated)) == (k_quad_special | k_quad_animated);
			animated[r] = decoded[is_animated ? min((marker & k_quad_ordinal_mask) - first_ordinal, decoded_quads_per_wave - 1) : 0u];
		}

		const uint32_t row = params.instance_rows != nullptr ? as_constant(params.instance_rows)[instance] : instance;
		f32x4* pose = reinterpret_cast<f32x4*>(poses + uint64_t(row) * pose_stride_bytes) + first_quad + lane;
		#pragma unroll
		for (uint32_t r = 0; r < k_rows; ++r)
		{
			f32x4 value = staged[r];
			const uint32_t marker = __float_as_uint(value.w);
			const bool is_special = int32_t(marker) < 0;
			const bool is_animated = is_special && (marker & k_quad_animated) != 0;
			// a default sub-track: the base pose holds the track_writer default's xyz, its W is 0 or 1
			value.w = is_special ? ((marker & k_quad_default_w_one) != 0 ? 1.0f : 0.0f) : value.w;
			value = is_animated ? animated[r] : value;
			if (r < full_rows || (r == full_rows && r * k_wave_size + lane < window_quads))
				store_streaming(&pose[r * k_wave_size], value);
		}
	}
