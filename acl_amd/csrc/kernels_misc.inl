// kernels_misc.inl -- part of aclhip.hip (one translation unit; included there, in this order, not compiled on its own).
// Small kernels: the sample list of convert_track_list, database tier metadata, the write bandwidth probe.

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

	// Measurement aid: streams `num_quads` float4 to HBM, 16 bytes per lane, to find the write bandwidth a pose-shaped store
	// stream can reach on this device (the decode kernel is a write streamer).
	__global__ __launch_bounds__(k_block_size) void stream_write_kernel(float4* __restrict__ destination, uint64_t num_quads, float seed)
	{
		const uint64_t stride = uint64_t(gridDim.x) * k_block_size;
		const float4 value = make_float4(seed, seed + 1.0f, seed + 2.0f, seed + 3.0f);
		for (uint64_t quad = uint64_t(blockIdx.x) * k_block_size + threadIdx.x; quad < num_quads; quad += stride)
			destination[quad] = value;
	}
