// kernels_misc.inl -- part of aclhip.hip (one translation unit; included there, in this order, not compiled on its own).
// Small kernels: the sample list of convert_track_list, database tier metadata, the device side locality order, the write bandwidth probe.

	// The instance list of convert_track_list's sampling loop (compression/impl/convert.impl.h:161-166): one instance per sample at
	// min(float(i) / sample_rate, duration), with the correctly rounded fp32 division the host code performs.
	__global__ void fill_sample_instances_kernel(uint32_t clip_id, uint32_t num_samples, float sample_rate, float duration, uint32_t* __restrict__ clip_ids, float* __restrict__ sample_times)
	{
		const uint32_t sample_index = blockIdx.x * blockDim.x + threadIdx.x;
		if (sample_index >= num_samples)
			return;
		clip_ids[sample_index] = clip_id;
		sample_times[sample_index] = fminf(float(sample_index) / sample_rate, duration);
	}

	// One entry per (chunk, segment) of a database tier: which runtime segment header the chunk's keyframes belong to and what
	// its tier metadata is while the chunk is resident ((samples_offset << 32) | sample_indices, database.impl.h:195-197).
	struct tier_patch
	{
		uint32_t segment_header_offset;		// into the runtime clip/segment header block
		uint32_t sample_indices;
		uint32_t samples_offset;
	};

	// Publishes (stream in) or retires (stream out) the tier metadata of a range of patches. Enqueued on the stream that carried
	// the bulk data copy, so a decode enqueued later on that stream sees both; decodes racing on other streams see either the old
	// or the new 64 bit value, like the reference's relaxed atomics (database_streamer.impl.h:108-110, database.impl.h:616-618).
	__global__ void apply_tier_metadata_kernel(uint8_t* __restrict__ runtime_headers, const tier_patch* __restrict__ patches, uint32_t first, uint32_t count, uint32_t tier_index, uint32_t stream_in)
	{
		const uint32_t index = blockIdx.x * blockDim.x + threadIdx.x;
		if (index >= count)
			return;
		const tier_patch patch = patches[first + index];
		unsigned long long* metadata = reinterpret_cast<unsigned long long*>(runtime_headers + patch.segment_header_offset) + tier_index;
		const unsigned long long value = stream_in != 0 ? ((static_cast<unsigned long long>(patch.samples_offset) << 32) | patch.sample_indices) : 0ull;
		__hip_atomic_store(metadata, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	}

	// Copies the runtime tier metadata of every segment into the sample records of the clips bound to the database
	// (database_sample_record, aclhip_device.h): one workgroup per bound clip, threads <-> samples. Enqueued behind
	// apply_tier_metadata_kernel by every stream_in / stream_out, and once for a clip when it is bound. `bound_clips` is the host's
	// list as of some moment between the call and the execution: an entry is acted on only if its table record (read now) is a
	// live clip bound to THIS database -- a handle that was unregistered or recycled in between is skipped.
	__global__ __launch_bounds__(256) void refresh_database_sample_tiers_kernel(const device_clip* __restrict__ clips, uint32_t num_clips,
		const uint32_t* __restrict__ bound_clips, uint32_t num_bound_clips, const uint8_t* __restrict__ runtime_headers)
	{
		if (blockIdx.x >= num_bound_clips)
			return;
		const uint32_t clip_id = bound_clips[blockIdx.x];
		if (clip_id >= num_clips)
			return;
		const device_clip clip = clips[clip_id];
		if ((clip.flags & (k_clip_valid | k_clip_database_samples)) != (k_clip_valid | k_clip_database_samples) || clip.db_headers != runtime_headers)
			return;
		database_sample_record* samples = reinterpret_cast<database_sample_record*>(const_cast<sample_record*>(clip.samples));
		const uint8_t* segment_headers = runtime_headers + clip.db_clip_header_offset + sizeof(database_runtime_clip_header);
		for (uint32_t sample = threadIdx.x; sample < clip.num_samples; sample += blockDim.x)
		{
			const uint32_t segment = samples[sample].record.segment_and_local >> 5;
			const unsigned long long* metadata = reinterpret_cast<const unsigned long long*>(segment_headers + sizeof(database_runtime_segment_header) * segment);
			for (uint32_t tier = 0; tier < 2; ++tier)
			{
				const unsigned long long value = __hip_atomic_load(metadata + tier, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				__hip_atomic_store(reinterpret_cast<unsigned long long*>(&samples[sample].tier_metadata[tier]), value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			}
		}
	}

	// Measurement aid: streams `num_quads` float4 to HBM, 16 bytes per lane, to find the write bandwidth a pose-shaped store
	// stream can reach on this device (the decode kernel is a write streamer).
	__global__ __launch_bounds__(k_block_size) void stream_write_kernel(float4* __restrict__ destination, uint64_t num_quads, float seed)
	{
		const uint64_t stride = uint64_t(gridDim.x) * k_block_size;
		const float4 value = make_float4(seed, seed + 1.0f, seed + 2.0f, seed + 3.0f);
		for (uint64_t quad = uint64_t(blockIdx.x) * k_block_size + threadIdx.x; quad < num_quads; quad += stride)
			destination[quad] = value;
	}

	// Measurement aid: what the write path sustains for the pose kernels' OWN store pattern -- one wave per pose window, 1 KiB streaming
	// stores to the window's rows, nothing else -- with the workgroups per CU limited by the LDS the launch asks for. The best of a few
	// occupancies is the ceiling bench.py quotes next to the 8 TB/s of the HBM3E specification: no kernel that writes these poses in
	// this pattern can be faster (6.0 - 6.7 TB/s: profiles/r03_experiments.md, tools/write_probe5.hip).
	// `chain_hops` dependent scalar loads in front of the stores pace the stream the way a decode's seek does: a write stream that is
	// paced reaches more than one issued as fast as waves can start, so the ceiling is the best over both knobs.
	__global__ __launch_bounds__(k_block_size) void pose_store_stream_kernel(uint8_t* __restrict__ poses, uint64_t pose_stride_bytes, uint32_t num_instances,
		uint32_t pose_quads, uint32_t windows_per_instance, const uint32_t* __restrict__ chain, uint32_t chain_hops, float seed)
	{
		extern __shared__ uint8_t occupancy_padding[];
		const uint32_t lane = threadIdx.x & (k_wave_size - 1);
		const uint32_t work_item = blockIdx.x * k_waves_per_block + __builtin_amdgcn_readfirstlane(threadIdx.x / k_wave_size);
		uint32_t hop = work_item & 4095u;
		for (uint32_t h = 0; h < chain_hops; ++h)
			hop = as_constant(chain)[hop & 4095u] + h;
		const uint32_t instance = work_item / windows_per_instance;
		const uint32_t first_quad = (work_item - instance * windows_per_instance) * k_image_chunk_quads;
		if (instance >= num_instances || first_quad >= pose_quads)
			return;
		const uint32_t window_quads = min(pose_quads - first_quad, k_image_chunk_quads);
		f32x4* pose = reinterpret_cast<f32x4*>(poses + uint64_t(instance) * pose_stride_bytes) + first_quad + lane;
		const f32x4 value = { seed, seed + 1.0f, float(hop), float(work_item) };
		constexpr uint32_t k_rows = (k_image_chunk_quads + k_wave_size - 1) / k_wave_size;
		#pragma unroll
		for (uint32_t r = 0; r < k_rows; ++r)
			if (r * k_wave_size + lane < window_quads)
				store_streaming(&pose[r * k_wave_size], value);
		if (seed == -1.0f && occupancy_padding[0] == 255)
			store_streaming(pose, value);		// keeps the LDS allocation alive
	}

	// ---- decode order for batches that draw on many clips (aclhip_order_instances_device; host twin in host_launch.inl) ----
	// Slot j of a launch (the j-th instance of the list) starts at wave j * windows_per_instance, workgroup b holds k_waves_per_block
	// consecutive waves and runs on XCD b % 8: which slots an XCD serves repeats with a period of at most 32 slots. The instances are
	// bucketed by clip into one sequence; XCD x serves positions [range_begin[x], range_begin[x + 1]) of it, through its own slots in
	// ascending order -- every clip is decoded on one XCD (two where a range boundary cuts it), next to its other instances.
	constexpr uint32_t k_num_xcds = 8;
	struct order_layout
	{
		uint32_t period;							// slots per period
		uint32_t per_xcd[k_num_xcds];				// slots of XCD x in a period
		uint32_t range_begin[k_num_xcds + 1];		// positions of the bucketed sequence served by XCD x
		uint8_t slots[k_num_xcds][32];				// XCD x's slots within a period, ascending
	};

	__host__ __device__ inline uint32_t order_slot_of(const order_layout& layout, uint32_t position)
	{
		uint32_t xcd = 0;
		for (uint32_t x = 1; x < k_num_xcds; ++x)
			xcd += position >= layout.range_begin[x] ? 1u : 0u;
		const uint32_t rank = position - layout.range_begin[xcd];
		const uint32_t per_period = layout.per_xcd[xcd];
		return (rank / per_period) * layout.period + layout.slots[xcd][rank % per_period];
	}

	// Device scope atomics on a handful of addresses (one per clip) run at the memory fabric's pace, serialized per address: a
	// workgroup first gathers its 2048 instances per clip in an LDS hash table (LDS atomics), then touches each of its clips' bins once.
#if !defined(ACLHIP_ORDER_BLOCK_SIZE)
	#define ACLHIP_ORDER_BLOCK_SIZE 1024
#endif
	constexpr uint32_t k_order_block_size = ACLHIP_ORDER_BLOCK_SIZE;
	#if !defined(ACLHIP_ORDER_INSTANCES_PER_THREAD)
#define ACLHIP_ORDER_INSTANCES_PER_THREAD 2
#endif
	constexpr uint32_t k_order_instances_per_thread = ACLHIP_ORDER_INSTANCES_PER_THREAD;
	constexpr uint32_t k_order_instances_per_block = k_order_block_size * k_order_instances_per_thread;
	constexpr uint32_t k_order_table_size = 2 * k_order_instances_per_block;		// load factor <= 0.5
	constexpr uint32_t k_order_empty_key = 0xFFFFFFFFu;								// never a bin: bins are clamped to num_bins - 1

	struct order_table
	{
		uint32_t keys[k_order_table_size];
		uint32_t counts[k_order_table_size];
	};

	__device__ inline void order_table_clear(order_table& table)
	{
		for (uint32_t slot = threadIdx.x; slot < k_order_table_size; slot += k_order_block_size)
		{
			table.keys[slot] = k_order_empty_key;
			table.counts[slot] = 0;
		}
		__syncthreads();
	}

	// the slot of `bin` in the workgroup's table and the instance's rank among the workgroup's instances of that bin
	__device__ inline uint32_t order_table_insert(order_table& table, uint32_t bin, uint32_t& rank)
	{
		uint32_t slot = ((bin * 2654435761u) >> 16) & (k_order_table_size - 1);
		for (;;)
		{
			const uint32_t found = atomicCAS(&table.keys[slot], k_order_empty_key, bin);
			if (found == k_order_empty_key || found == bin)
				break;
			slot = (slot + 1) & (k_order_table_size - 1);
		}
		rank = atomicAdd(&table.counts[slot], 1u);
		return slot;
	}

	// instances per clip; handles past the registry (the decode rejects them) share the last bin
	__global__ __launch_bounds__(k_order_block_size) void order_count_kernel(const uint32_t* __restrict__ clip_ids, uint32_t num_instances, uint32_t num_bins, uint32_t* __restrict__ bins)
	{
		__shared__ order_table table;
		order_table_clear(table);
		for (uint32_t k = 0; k < k_order_instances_per_thread; ++k)
		{
			const uint32_t instance = blockIdx.x * k_order_instances_per_block + k * k_order_block_size + threadIdx.x;
			uint32_t rank;
			if (instance < num_instances)
				order_table_insert(table, min(clip_ids[instance], num_bins - 1), rank);
		}
		__syncthreads();
		for (uint32_t slot = threadIdx.x; slot < k_order_table_size; slot += k_order_block_size)
			if (table.keys[slot] != k_order_empty_key)
				atomicAdd(&bins[table.keys[slot]], table.counts[slot]);
	}

	// counts -> first position of every clip (the cursors of the scatter); leaves the counters at zero for the next call. One
	// workgroup, 4096 bins per iteration (the arrays are padded to that)
	__global__ __launch_bounds__(1024) void order_scan_kernel(uint32_t* __restrict__ counters, uint32_t* __restrict__ cursors, uint32_t num_bins)
	{
		__shared__ uint32_t wave_totals[16];
		__shared__ uint32_t carry;
		const uint32_t lane = threadIdx.x & (k_wave_size - 1);
		const uint32_t wave = threadIdx.x / k_wave_size;
		if (threadIdx.x == 0)
			carry = 0;
		__syncthreads();
		for (uint32_t base = 0; base < num_bins; base += 4096)
		{
			uint4* quad = reinterpret_cast<uint4*>(counters + base) + threadIdx.x;
			const uint4 counts = *quad;
			*quad = uint4{ 0, 0, 0, 0 };
			const uint32_t sum = counts.x + counts.y + counts.z + counts.w;
			uint32_t inclusive = sum;
			for (uint32_t step = 1; step < k_wave_size; step *= 2)
			{
				const uint32_t below = __shfl_up(inclusive, step);
				if (lane >= step)
					inclusive += below;
			}
			if (lane == k_wave_size - 1)
				wave_totals[wave] = inclusive;
			__syncthreads();
			uint32_t first = carry + inclusive - sum;
			for (uint32_t w = 0; w < wave; ++w)
				first += wave_totals[w];
			reinterpret_cast<uint4*>(cursors + base)[threadIdx.x] = uint4{ first, first + counts.x, first + counts.x + counts.y, first + counts.x + counts.y + counts.z };
			__syncthreads();
			if (threadIdx.x == 1023)
				carry = first + sum;
			__syncthreads();
		}
	}

	// every instance takes the next position of its clip and lands in the slot that position maps to
	__global__ __launch_bounds__(k_order_block_size) void order_scatter_kernel(const uint32_t* __restrict__ clip_ids, const float* __restrict__ sample_times, uint32_t num_instances,
		uint32_t num_bins, uint32_t* __restrict__ bins, order_layout layout_argument, uint32_t* __restrict__ out_order, uint32_t* __restrict__ out_clip_ids, float* __restrict__ out_sample_times,
		uint32_t* __restrict__ out_positions, uint32_t* __restrict__ host_failed)
	{
		__shared__ order_table table;
		__shared__ order_layout layout;
		if (threadIdx.x < sizeof(order_layout) / 4)
			reinterpret_cast<uint32_t*>(&layout)[threadIdx.x] = reinterpret_cast<const uint32_t*>(&layout_argument)[threadIdx.x];
		order_table_clear(table);

		uint32_t clip_id[k_order_instances_per_thread], slot[k_order_instances_per_thread], rank[k_order_instances_per_thread];
		for (uint32_t k = 0; k < k_order_instances_per_thread; ++k)
		{
			const uint32_t instance = blockIdx.x * k_order_instances_per_block + k * k_order_block_size + threadIdx.x;
			if (instance < num_instances)
			{
				clip_id[k] = clip_ids[instance];
				slot[k] = order_table_insert(table, min(clip_id[k], num_bins - 1), rank[k]);
			}
		}
		__syncthreads();
		// the workgroup's instances of a clip take consecutive positions: counts[] becomes the first of them
		for (uint32_t entry = threadIdx.x; entry < k_order_table_size; entry += k_order_block_size)
			if (table.keys[entry] != k_order_empty_key)
				table.counts[entry] = atomicAdd(&bins[table.keys[entry]], table.counts[entry]);
		__syncthreads();
		for (uint32_t k = 0; k < k_order_instances_per_thread; ++k)
		{
			const uint32_t instance = blockIdx.x * k_order_instances_per_block + k * k_order_block_size + threadIdx.x;
			if (instance >= num_instances)
				continue;
			const uint32_t destination = order_slot_of(layout, table.counts[slot[k]] + rank[k]);
			// (a position past the list: this launch's counters were written by ANOTHER launch at the same time -- a captured ordering replayed
			// on another stream than the one whose scratch it holds, next to an ordering of that stream. Nothing is written out of bounds, and
			// the host hears about it like about a barrier that gave up: the order this launch leaves is not a permutation.)
			if (destination >= num_instances)
			{
				if (host_failed != nullptr)
					__hip_atomic_store(host_failed, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
				continue;
			}
			out_order[destination] = instance;
			if (out_clip_ids != nullptr)
				out_clip_ids[destination] = clip_id[k];
			if (out_sample_times != nullptr)
				out_sample_times[destination] = sample_times[instance];
			if (out_positions != nullptr)
				out_positions[instance] = destination;
		}
	}

	// Instance lists kept in decode order (aclhip_instance_list_update): instance instances[k] now plays clips[k]. It keeps its slot --
	// a full re-order follows once enough of the list has changed (host_lists.inl) --, the clip handle the decode reads for that slot changes.
	// (Two entries for one instance race on both arrays independently: the header forbids them. Handles are not checked: the decode does.)
	__global__ __launch_bounds__(256) void update_instance_list_kernel(const uint32_t* __restrict__ instances, const uint32_t* __restrict__ new_clips, uint32_t count,
		uint32_t num_instances, uint32_t* __restrict__ list_clips, const uint32_t* __restrict__ positions, uint32_t* __restrict__ ordered_clips)
	{
		const uint32_t index = blockIdx.x * blockDim.x + threadIdx.x;
		if (index >= count)
			return;
		const uint32_t instance = instances[index];
		if (instance >= num_instances)
			return;
		list_clips[instance] = new_clips[index];
		ordered_clips[positions[instance]] = new_clips[index];
	}

	// ---- the same order without device scope atomics on the bins: per workgroup histograms laid out as a matrix, ONE launch ----
	// (Device scope atomics on a few hundred hot addresses are what the three kernels above spend their time on: 23 us for 64k
	// instances; three dependent launches cost ~5 us each before they do any work.) While the clip table has at most
	// k_order_direct_bins entries, every workgroup counts its instances per clip in a directly indexed LDS histogram and writes its
	// column of a [bin][workgroup] matrix; the workgroups meet at a barrier in global memory; every workgroup turns its share of the
	// rows into prefix sums (in place) and row totals; a second barrier; every workgroup derives its own first positions from the
	// totals and its column and places its instances. The only device scope atomics are the barriers': one add per workgroup and
	// barrier, one polled word. Nothing to zero between calls.
	// The workgroups wait for one another: the host launches this form with at most 64 workgroups and never more than the device
	// has compute units (all of them become resident as soon as whatever else runs on the device drains).
	// Measured (64k instances, 257 bins, back to back): see DESIGN 4.10; the same matrix in three launches (histogram | one workgroup
	// scans the matrix | place) 16.8 us; one launch where every workgroup reads the whole matrix behind ONE barrier 15.4 - 20.6 us.
	constexpr uint32_t k_order_direct_bins = 8192;			// 32 KB of LDS
	constexpr uint32_t k_order_direct_block_size = 1024;

	struct order_control
	{
		uint32_t arrived;		// workgroups whose column is written (back to 0 when the last one has arrived)
		uint32_t padding[31];	// (the polls of the word below must not queue in front of the adds to the word above)
		uint32_t generation;	// barriers passed: every workgroup reads it when it starts and waits for the next value (nothing
								// a captured hipGraph would have to change between replays)
		uint32_t more_padding[31];
		uint32_t failed;		// != 0: a barrier of some launch did not open (see order_grid_barrier); the host resets the words and stops using this form on the stream
		uint32_t last_padding[31];
	};
	constexpr uint32_t k_order_grid_entries = 1u << 19;			// bins x workgroups the one launch form takes
	constexpr uint32_t k_order_grid_max_log2_blocks = 6;
#if !defined(ACLHIP_ORDER_POLL_SLEEP)
	#define ACLHIP_ORDER_POLL_SLEEP 4
#endif
	constexpr uint32_t k_order_barrier_poll_sleep = ACLHIP_ORDER_POLL_SLEEP;		// x 64 clocks between two polls
	constexpr uint32_t k_order_barrier_max_polls = 1u << 22;	// seconds

	// All workgroups of the grid have arrived, and what they wrote before is visible. `generation`: thread 0's, the value the barrier's
	// word has to leave. The XCDs' L2s are not coherent with one another: an arriving workgroup RELEASES at agent scope (its L2 writes
	// back), a passing one ACQUIRES (its L2 forgets what other XCDs own) -- one thread per workgroup does both, once, behind a
	// workgroup barrier in front of which every wave has waited for its own stores; the polls in between are relaxed loads.
	// (Round 3 first shipped this with agent scope ACCESSES and no fences -- 1.5 us faster, and right on an idle device; with sixteen
	// orderings in flight at once, tools/order_stress.py caught one wrong order in 640 000: an acknowledged write-through is not yet
	// a visible one.)
	// A barrier that does not open within seconds cannot open any more -- a fault in an earlier call left its words behind, or so many
	// ordering launches of OTHER processes share the device that none of them gets all its workgroups resident (the grid is sized from
	// the occupancy query to fit an otherwise idle device, order_instances_on_device). Round 3 TRAPPED there, which takes the whole
	// process down with the queue. Now the workgroup gives up: it raises `failed` (device word) and `*host_failed` (pinned host memory
	// the host looks at in its next ordering / list call on the stream, without synchronizing), and so does every other workgroup of
	// the launch: each writes the identity order for its share of the instances (order_write_identity below). The host then reports
	// the failure, resets the barrier words and orders with the three launch form (no workgroup of which waits for another) on that
	// stream from then on.
	constexpr uint32_t k_order_generation_given_up = 0x80000000u;	// in control->generation: some workgroup gave up on this barrier, nobody passes it any more

	__device__ __forceinline__ bool order_grid_barrier(order_control* control, uint32_t* host_failed, uint32_t generation, uint32_t max_polls, uint32_t& passed)
	{
		__builtin_amdgcn_s_waitcnt(0);		// (vmcnt 0: this wave's stores have reached the L2)
		__syncthreads();
		if (threadIdx.x == 0)
		{
			// The barrier opens for ALL workgroups or for none: opening it and giving up on it are both a compare-and-swap of the
			// generation word away from `generation`, and only one of them can win (until round 5 a barrier that opened in the very
			// poll in which another workgroup gave up left an order that was half placed, half not).
			uint32_t open;
			if (__hip_atomic_fetch_add(&control->arrived, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1)
			{
				__hip_atomic_store(&control->arrived, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				__builtin_amdgcn_s_waitcnt(0);		// (arrived is back to 0 before anybody passes)
				uint32_t expected = generation;
				open = __hip_atomic_compare_exchange_strong(&control->generation, &expected, (generation + 1u) & ~k_order_generation_given_up, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) ? 1u : 0u;
			}
			else
			{
				uint32_t polls = 0;
				uint32_t seen = generation;
				while ((seen = __hip_atomic_load(&control->generation, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == generation && ++polls < max_polls)
					__builtin_amdgcn_s_sleep(k_order_barrier_poll_sleep);
				if (seen == generation)
				{
					// out of patience: close the barrier for everybody -- unless it opened this very moment
					uint32_t expected = generation;
					if (__hip_atomic_compare_exchange_strong(&control->generation, &expected, generation | k_order_generation_given_up, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
						seen = generation | k_order_generation_given_up;
					else
						seen = expected;
				}
				open = (seen & k_order_generation_given_up) == 0 ? 1u : 0u;
			}
			if (open == 0)
			{
				__hip_atomic_store(&control->failed, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				__hip_atomic_store(host_failed, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
			}
			__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
			passed = open;
		}
		__syncthreads();
		return passed != 0;
	}

	// What a workgroup that gives up leaves behind for its share of the instances: the IDENTITY order -- slot i holds instance i. Every
	// workgroup of a launch gives up or none does (order_grid_barrier), so a failed ordering is still a valid one, without locality:
	// a decode that was enqueued behind it reads and writes in bounds, and an instance list stays consistent (order, positions and
	// ordered clips agree). The host still reports the failure at its next call on the stream.
	__device__ __forceinline__ void order_write_identity(const uint32_t* __restrict__ clip_ids, const float* __restrict__ sample_times, uint32_t first, uint32_t end,
		uint32_t* __restrict__ out_order, uint32_t* __restrict__ out_clip_ids, float* __restrict__ out_sample_times, uint32_t* __restrict__ out_positions)
	{
		for (uint32_t instance = first + threadIdx.x; instance < end; instance += blockDim.x)
		{
			out_order[instance] = instance;
			if (out_clip_ids != nullptr)
				out_clip_ids[instance] = clip_ids[instance];
			if (out_sample_times != nullptr)
				out_sample_times[instance] = sample_times[instance];
			if (out_positions != nullptr)
				out_positions[instance] = instance;
		}
	}

#if defined(ACLHIP_EXPERIMENTS)
	// wall clock stamps (10 ns units) of workgroup phases: tools/order_phases.py
	__device__ unsigned long long g_order_stamps[64 * 8];
	#define ACLHIP_ORDER_STAMP(index) do { if (threadIdx.x == 0 && blockIdx.x < 64) g_order_stamps[blockIdx.x * 8 + (index)] = wall_clock64(); } while (false)
#else
	#define ACLHIP_ORDER_STAMP(index) do { } while (false)
#endif

	__global__ __launch_bounds__(k_order_direct_block_size) void order_instances_grid_kernel(const uint32_t* __restrict__ clip_ids, const float* __restrict__ sample_times, uint32_t num_instances,
		uint32_t instances_per_block, uint32_t num_bins, uint32_t log2_blocks, uint32_t* histograms, order_control* control, uint32_t* host_failed, uint32_t max_polls, uint32_t absent_block, order_layout layout_argument,
		uint32_t* __restrict__ out_order, uint32_t* __restrict__ out_clip_ids, float* __restrict__ out_sample_times, uint32_t* __restrict__ out_positions)
	{
		__shared__ uint32_t cursors[k_order_direct_bins];		// instances per bin of this workgroup, then their first positions
		__shared__ order_layout layout;
		__shared__ uint32_t wave_totals[k_order_direct_block_size / k_wave_size];
		__shared__ uint32_t passed;
		const uint32_t lane = threadIdx.x & (k_wave_size - 1);
		const uint32_t wave = threadIdx.x / k_wave_size;
		ACLHIP_ORDER_STAMP(0);
		uint32_t generation = 0;
		if (threadIdx.x == 0)
		{
			generation = __hip_atomic_load(&control->generation, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);	// (before this workgroup arrives: the barrier cannot have opened yet)
			// the words an earlier, failed launch left behind cannot be trusted (a replayed hipGraph runs before the host has seen the failure)
			passed = __hip_atomic_load(&control->failed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0 && (generation & k_order_generation_given_up) == 0 ? 1u : 0u;
		}
		__syncthreads();
		const uint32_t first = blockIdx.x * instances_per_block, end = min(first + instances_per_block, num_instances);
		// (absent_block: a test's stand-in for a workgroup that becomes resident too late -- tests/test_gpu_order_device.py; no block has
		// this index otherwise. By the time such a workgroup runs the others have given up: it finds `failed` raised and does what they did)
		if (passed == 0 || blockIdx.x == absent_block)
		{
			order_write_identity(clip_ids, sample_times, first, end, out_order, out_clip_ids, out_sample_times, out_positions);
			return;
		}
		if (threadIdx.x < sizeof(order_layout) / 4)
			reinterpret_cast<uint32_t*>(&layout)[threadIdx.x] = reinterpret_cast<const uint32_t*>(&layout_argument)[threadIdx.x];
		for (uint32_t bin = threadIdx.x; bin < num_bins; bin += k_order_direct_block_size)
			cursors[bin] = 0;
		__syncthreads();
		for (uint32_t instance = first + threadIdx.x; instance < end; instance += k_order_direct_block_size)
			atomicAdd(&cursors[min(clip_ids[instance], num_bins - 1)], 1u);
		__syncthreads();
		for (uint32_t bin = threadIdx.x; bin < num_bins; bin += k_order_direct_block_size)
			histograms[(size_t(bin) << log2_blocks) + blockIdx.x] = cursors[bin];

		// every column is written and visible
		ACLHIP_ORDER_STAMP(1);
		if (!order_grid_barrier(control, host_failed, generation, max_polls, passed))
		{
			order_write_identity(clip_ids, sample_times, first, end, out_order, out_clip_ids, out_sample_times, out_positions);
			return;
		}
		ACLHIP_ORDER_STAMP(3);

		// Every workgroup turns the rows of ITS share of the bins into "instances in the workgroups in front" (in place) and the
		// bin's total (behind the matrix): a bin's 1 << log2_blocks entries lie in consecutive lanes of one wave. (Every workgroup
		// reading the whole matrix instead: 14 us of agent scope loads for 64 workgroups x 257 bins.)
		const uint32_t num_blocks = 1u << log2_blocks;
		uint32_t* bin_totals = histograms + (size_t(num_bins) << log2_blocks);
		{
			const uint32_t bins_per_block = (num_bins + num_blocks - 1) >> log2_blocks;
			const uint32_t first_bin = blockIdx.x * bins_per_block, end_bin = min(first_bin + bins_per_block, num_bins);
			const uint32_t first_entry = first_bin << log2_blocks, end_entry = end_bin << log2_blocks;
			const uint32_t lane_in_row = lane & (num_blocks - 1);
			for (uint32_t base = first_entry; base < end_entry; base += k_order_direct_block_size)
			{
				const uint32_t entry = base + threadIdx.x;
				const uint32_t count = entry < end_entry ? histograms[entry] : 0u;
				uint32_t inclusive = count;
				for (uint32_t step = 1; step < num_blocks; step *= 2)
				{
					const uint32_t lower = __shfl_up(inclusive, step);
					if (lane_in_row >= step)
						inclusive += lower;
				}
				if (entry < end_entry)
				{
					histograms[entry] = inclusive - count;
					if (lane_in_row == num_blocks - 1)
						bin_totals[entry >> log2_blocks] = inclusive;
				}
			}
		}
		ACLHIP_ORDER_STAMP(2);
		if (!order_grid_barrier(control, host_failed, (generation + 1u) & ~k_order_generation_given_up, max_polls, passed))
		{
			order_write_identity(clip_ids, sample_times, first, end, out_order, out_clip_ids, out_sample_times, out_positions);
			return;
		}
		ACLHIP_ORDER_STAMP(4);

		// first position of a bin = instances of the bins in front of it; this workgroup starts behind the ones in front of it.
		// Consecutive bins per thread, as few as cover the table (the loads are agent scope: a microsecond, whatever their number)
		constexpr uint32_t max_bins_per_thread = k_order_direct_bins / k_order_direct_block_size;
		{
			const uint32_t bins_per_thread = (num_bins + k_order_direct_block_size - 1) / k_order_direct_block_size;
			const uint32_t first_bin = threadIdx.x * bins_per_thread;
			uint32_t counts[max_bins_per_thread], in_front[max_bins_per_thread];
			#pragma unroll
			for (uint32_t k = 0; k < max_bins_per_thread; ++k)
			{
				const bool valid = k < bins_per_thread && first_bin + k < num_bins;
				counts[k] = valid ? bin_totals[first_bin + k] : 0u;
				in_front[k] = valid ? histograms[(size_t(first_bin + k) << log2_blocks) + blockIdx.x] : 0u;
			}
			uint32_t sum = 0;
			#pragma unroll
			for (uint32_t k = 0; k < max_bins_per_thread; ++k)
				sum += counts[k];
			uint32_t inclusive = sum;
			for (uint32_t step = 1; step < k_wave_size; step *= 2)
			{
				const uint32_t lower = __shfl_up(inclusive, step);
				if (lane >= step)
					inclusive += lower;
			}
			if (lane == k_wave_size - 1)
				wave_totals[wave] = inclusive;
			__syncthreads();
			uint32_t position = inclusive - sum;
			for (uint32_t w = 0; w < wave; ++w)
				position += wave_totals[w];
			#pragma unroll
			for (uint32_t k = 0; k < max_bins_per_thread; ++k)
			{
				if (k < bins_per_thread && first_bin + k < num_bins)
					cursors[first_bin + k] = position + in_front[k];
				position += counts[k];
			}
		}
		__syncthreads();
		ACLHIP_ORDER_STAMP(5);

		for (uint32_t instance = first + threadIdx.x; instance < end; instance += k_order_direct_block_size)
		{
			const uint32_t clip_id = clip_ids[instance];
			const uint32_t destination = order_slot_of(layout, atomicAdd(&cursors[min(clip_id, num_bins - 1)], 1u));
			// A position past the list: the matrix or the barrier words of this launch were written by ANOTHER launch at the same time --
			// a captured ordering replayed on another stream than the one whose scratch it holds, next to an ordering of that stream
			// (include/aclhip.h: aclhip_order_instances_device). Nothing is written out of bounds; the failure is raised like a barrier's.
			if (destination >= num_instances)
			{
				__hip_atomic_store(&control->failed, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				__hip_atomic_store(host_failed, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
				continue;
			}
			out_order[destination] = instance;
			if (out_clip_ids != nullptr)
				out_clip_ids[destination] = clip_id;
			if (out_sample_times != nullptr)
				out_sample_times[destination] = sample_times[instance];
			if (out_positions != nullptr)
				out_positions[instance] = destination;
		}
		ACLHIP_ORDER_STAMP(6);
#if defined(ACLHIP_EXPERIMENTS)
		__builtin_amdgcn_s_waitcnt(0);		// (vmcnt 0: the stores acknowledged)
		__syncthreads();
		ACLHIP_ORDER_STAMP(7);
#endif
	}
