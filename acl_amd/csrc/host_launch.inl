// host_launch.inl -- part of aclhip.hip (one translation unit; included there, in this order, not compiled on its own).
// Host side: pose and single track launches, locality order of instance lists.

namespace
{
	aclhip_status launch_tracks(aclhip_context* context, const aclhip_clip* clips, const float* sample_times, uint32_t num_instances,
		const decode_params& params, void* poses, uint64_t pose_stride_bytes, hipStream_t stream)
	{
		// launches read the registry's maxima and are noted (note_launch_stream) under the registry lock; the clip table itself never moves
		std::lock_guard<std::mutex> lock(context->mutex);
		note_launch_stream(context, stream);

		// one wave per (instance, pose window); instances of clips with fewer windows than the largest registered clip leave waves idle
		const uint32_t windows_per_instance = std::max<uint32_t>((context->max_pose_quads + k_image_chunk_quads - 1) / k_image_chunk_quads, 1);
		const uint64_t num_waves = uint64_t(num_instances) * windows_per_instance;
		if (num_waves > 0xFFFFFFFFull - k_waves_per_block)
			return fail(context, ACLHIP_ERROR_INVALID_ARGUMENT, "batch too large: %u instances x %u pose windows", num_instances, windows_per_instance);
		const uint32_t num_blocks = uint32_t((num_waves + k_waves_per_block - 1) / k_waves_per_block);

		// the common case (track_writer defaults, no per track rounding, normalization != always) copies a resolved pose image
		const bool any_settings = params.standard_defaults == 0 || params.per_track_rounding != 0 || context->force_generic_kernel;
		const uint32_t lds_quads_per_wave = std::min<uint32_t>(std::max<uint32_t>(align_to_u32(context->max_pose_quads, 64), 64), k_image_chunk_quads);
		size_t lds_bytes = size_t(lds_quads_per_wave) * 16 * k_waves_per_block;
		{
			// measurement aid: ACLHIP_EXTRA_LDS_BYTES inflates a workgroup's LDS so that fewer workgroups fit a CU (occupancy experiments)
			static const size_t extra_lds = []() { const char* value = std::getenv("ACLHIP_EXTRA_LDS_BYTES"); return value != nullptr ? size_t(std::atol(value)) : size_t(0); }();
			lds_bytes += extra_lds;
		}
		// an output descriptor that changes nothing (QVV48, nothing skipped) takes the plain kernels
		const bool compact = params.layout != ACLHIP_LAYOUT_QVV48 || params.skip_mask != 0;
		// the compact layouts with nothing else skipped have kernels that build the LDS image in the output layout (kernels_pose.inl)
		const bool native_layout = !any_settings && params.layout != ACLHIP_LAYOUT_QVV48 && (params.skip_mask & ~(params.layout == ACLHIP_LAYOUT_QV32 ? 4u : 0u)) == 0;
		const auto kernel = native_layout ? (params.layout == ACLHIP_LAYOUT_QV32 ? decompress_tracks_qv32_kernel : decompress_tracks_qvv40_kernel)
			: any_settings ? (compact ? decompress_tracks_any_settings_compact_kernel : decompress_tracks_any_settings_kernel)
			: (compact ? decompress_tracks_compact_kernel : decompress_tracks_kernel);
		hipLaunchKernelGGL(kernel, dim3(num_blocks), dim3(k_block_size), lds_bytes, stream,
			context->d_clips, context->d_clips_capacity, clips, sample_times, num_instances, windows_per_instance, params,
			static_cast<uint8_t*>(poses), pose_stride_bytes, lds_quads_per_wave, context->d_rejected);
		ACLHIP_CHECK_HIP(context, hipGetLastError());
		return ACLHIP_OK;
	}

	aclhip_status check_batch_arguments(aclhip_context* context, const void* clips, const void* sample_times, uint32_t num_instances, const void* out, uint64_t pose_stride_bytes)
	{
		if (context == nullptr)
			return ACLHIP_ERROR_INVALID_ARGUMENT;
		if (num_instances != 0 && (clips == nullptr || sample_times == nullptr || out == nullptr))
			return fail(context, ACLHIP_ERROR_INVALID_ARGUMENT, "null instance list or output buffer");
		if ((pose_stride_bytes & 15u) != 0 || (reinterpret_cast<uintptr_t>(out) & 15u) != 0)
			return fail(context, ACLHIP_ERROR_INVALID_ARGUMENT, "pose buffer and stride must be 16 byte aligned");
		return ACLHIP_OK;
	}
}

extern "C" aclhip_status aclhip_decompress_tracks_batch(aclhip_context* context, const aclhip_clip* clips, const float* sample_times, uint32_t num_instances,
	const aclhip_decompress_params* params, void* poses, uint64_t pose_stride_bytes, void* stream)
{
	aclhip_status status = check_batch_arguments(context, clips, sample_times, num_instances, poses, pose_stride_bytes);
	if (status != ACLHIP_OK)
		return status;
	if (num_instances == 0)
		return ACLHIP_OK;

	decode_params device_params;
	status = resolve_params(context, params, device_params);
	if (status != ACLHIP_OK)
		return status;

	device_guard guard(context->device);
	return launch_tracks(context, clips, sample_times, num_instances, device_params, poses, pose_stride_bytes, static_cast<hipStream_t>(stream));
}

extern "C" aclhip_status aclhip_decompress_tracks_batch_rows(aclhip_context* context, const aclhip_clip* clips, const float* sample_times, const uint32_t* rows,
	uint32_t num_instances, const aclhip_decompress_params* params, void* poses, uint64_t pose_stride_bytes, void* stream)
{
	aclhip_status status = check_batch_arguments(context, clips, sample_times, num_instances, poses, pose_stride_bytes);
	if (status != ACLHIP_OK)
		return status;
	if (num_instances == 0)
		return ACLHIP_OK;

	decode_params device_params;
	status = resolve_params(context, params, device_params);
	if (status != ACLHIP_OK)
		return status;
	device_params.instance_rows = rows;

	device_guard guard(context->device);
	return launch_tracks(context, clips, sample_times, num_instances, device_params, poses, pose_stride_bytes, static_cast<hipStream_t>(stream));
}

namespace
{
	// aclhip_output_desc -> decode_params
	aclhip_status apply_output_desc(aclhip_context* context, const aclhip_output_desc* output, decode_params& params)
	{
		if (output == nullptr)
			return ACLHIP_OK;
		if (output->layout > ACLHIP_LAYOUT_QV32)
			return fail(context, ACLHIP_ERROR_INVALID_ARGUMENT, "unknown pose layout %u", output->layout);
		params.layout = uint8_t(output->layout);
		params.skip_mask = uint8_t((output->skip_rotations != 0 ? 1u : 0u) | (output->skip_translations != 0 ? 2u : 0u) | (output->skip_scales != 0 ? 4u : 0u));
		if (output->layout == ACLHIP_LAYOUT_QV32)
			params.skip_mask |= 4u;
		params.instance_rows = output->rows;
		return ACLHIP_OK;
	}
}

extern "C" uint32_t aclhip_layout_bytes_per_track(uint32_t layout)
{
	return layout == ACLHIP_LAYOUT_QVV48 ? 48u : (layout == ACLHIP_LAYOUT_QVV40 ? 40u : (layout == ACLHIP_LAYOUT_QV32 ? 32u : 0u));
}

extern "C" aclhip_status aclhip_decompress_tracks_batch_out(aclhip_context* context, const aclhip_clip* clips, const float* sample_times, uint32_t num_instances,
	const aclhip_decompress_params* params, const aclhip_output_desc* output, void* poses, uint64_t pose_stride_bytes, void* stream)
{
	aclhip_status status = check_batch_arguments(context, clips, sample_times, num_instances, poses, pose_stride_bytes);
	if (status != ACLHIP_OK)
		return status;
	if (num_instances == 0)
		return ACLHIP_OK;

	decode_params device_params;
	status = resolve_params(context, params, device_params);
	if (status == ACLHIP_OK)
		status = apply_output_desc(context, output, device_params);
	if (status != ACLHIP_OK)
		return status;

	device_guard guard(context->device);
	return launch_tracks(context, clips, sample_times, num_instances, device_params, poses, pose_stride_bytes, static_cast<hipStream_t>(stream));
}

// Work order for batches that draw on many clips. Workgroup b of a launch runs on XCD b % 8 (each XCD has its own 4 MB L2) and
// holds k_waves_per_block consecutive (instance, pose window) work items: dealing the instances out so that every clip is only
// ever decoded on ONE XCD, next to its other instances, leaves each L2 with an eighth of the clips to keep.
extern "C" aclhip_status aclhip_order_instances_for_locality(const aclhip_context* context, const aclhip_clip* clips, uint32_t num_instances, uint32_t* out_order)
{
	if ((clips == nullptr || out_order == nullptr) && num_instances != 0)
		return ACLHIP_ERROR_INVALID_ARGUMENT;

	constexpr uint32_t k_num_xcds = 8;
	uint32_t windows_per_instance = 1;
	if (context != nullptr)
	{
		std::lock_guard<std::mutex> lock(const_cast<aclhip_context*>(context)->mutex);
		windows_per_instance = std::max<uint32_t>((context->max_pose_quads + k_image_chunk_quads - 1) / k_image_chunk_quads, 1);
	}
	// instances per workgroup; poses of several windows fill whole workgroups on their own, only the clip order matters then
	const uint32_t group = std::max<uint32_t>(k_waves_per_block / windows_per_instance, 1);

	return guarded(const_cast<aclhip_context*>(context), [&]() -> aclhip_status
	{
		// per XCD: its instances, bucketed by clip (stable: instances of a clip keep their relative order). A counting sort when the
		// handles are small numbers (they are slots of the registry), a comparison sort for arbitrary values
		std::vector<uint32_t> sorted(num_instances);
		uint32_t max_clip = 0;
		for (uint32_t i = 0; i < num_instances; ++i)
			max_clip = std::max(max_clip, clips[i]);
		if (uint64_t(max_clip) < uint64_t(num_instances) * 4 + 1024)
		{
			std::vector<uint32_t> position(size_t(max_clip) + 2, 0);
			for (uint32_t i = 0; i < num_instances; ++i)
				position[clips[i]]++;
			// first position of every clip in (clip % 8, clip) order
			uint32_t next = 0;
			for (uint32_t xcd = 0; xcd < k_num_xcds; ++xcd)
				for (uint64_t clip = xcd; clip <= max_clip; clip += k_num_xcds)
				{
					const uint32_t count = position[clip];
					position[clip] = next;
					next += count;
				}
			for (uint32_t i = 0; i < num_instances; ++i)
				sorted[position[clips[i]]++] = i;
		}
		else
		{
			for (uint32_t i = 0; i < num_instances; ++i)
				sorted[i] = i;
			std::stable_sort(sorted.begin(), sorted.end(), [&](uint32_t a, uint32_t b)
			{
				const uint32_t xcd_a = clips[a] % k_num_xcds, xcd_b = clips[b] % k_num_xcds;
				return xcd_a != xcd_b ? xcd_a < xcd_b : clips[a] < clips[b];
			});
		}
		uint32_t list_begin[k_num_xcds + 1] = {};
		for (uint32_t i = 0; i < num_instances; ++i)
			list_begin[clips[sorted[i]] % k_num_xcds + 1]++;
		for (uint32_t x = 0; x < k_num_xcds; ++x)
			list_begin[x + 1] += list_begin[x];

		// deal whole workgroups out round robin; an XCD whose list runs dry takes from the longest remaining list
		uint32_t cursor[k_num_xcds];
		for (uint32_t x = 0; x < k_num_xcds; ++x)
			cursor[x] = list_begin[x];
		uint32_t written = 0;
		for (uint32_t workgroup = 0; written < num_instances; ++workgroup)
		{
			uint32_t source = workgroup % k_num_xcds;
			if (cursor[source] == list_begin[source + 1])
			{
				uint32_t longest = 0;
				for (uint32_t x = 0; x < k_num_xcds; ++x)
					if (list_begin[x + 1] - cursor[x] > longest)
					{
						longest = list_begin[x + 1] - cursor[x];
						source = x;
					}
			}
			const uint32_t take = std::min<uint32_t>(group, list_begin[source + 1] - cursor[source]);
			for (uint32_t k = 0; k < take; ++k)
				out_order[written++] = sorted[cursor[source]++];
			// a short tail would shift every later workgroup's XCD: pad it from the longest list
			for (uint32_t k = take; k < group && written < num_instances; ++k)
			{
				uint32_t longest = 0, from = 0;
				for (uint32_t x = 0; x < k_num_xcds; ++x)
					if (list_begin[x + 1] - cursor[x] > longest)
					{
						longest = list_begin[x + 1] - cursor[x];
						from = x;
					}
				out_order[written++] = sorted[cursor[from]++];
			}
		}
		return ACLHIP_OK;
	});
}

extern "C" aclhip_status aclhip_decompress_track_batch(aclhip_context* context, const aclhip_clip* clips, const float* sample_times, const uint32_t* track_indices,
	uint32_t num_instances, const aclhip_decompress_params* params, void* transforms, void* stream)
{
	aclhip_status status = check_batch_arguments(context, clips, sample_times, num_instances, transforms, 48);
	if (status != ACLHIP_OK)
		return status;
	if (num_instances == 0)
		return ACLHIP_OK;
	if (track_indices == nullptr)
		return fail(context, ACLHIP_ERROR_INVALID_ARGUMENT, "null track index list");

	decode_params device_params;
	status = resolve_params(context, params, device_params);
	if (status != ACLHIP_OK)
		return status;

	std::lock_guard<std::mutex> lock(context->mutex);		// see launch_tracks
	device_guard guard(context->device);
	note_launch_stream(context, static_cast<hipStream_t>(stream));
	const uint32_t num_blocks = (num_instances + k_block_size - 1) / k_block_size;
	hipLaunchKernelGGL(decompress_track_kernel, dim3(num_blocks), dim3(k_block_size), 0, static_cast<hipStream_t>(stream),
		context->d_clips, context->d_clips_capacity, clips, sample_times, track_indices, num_instances, device_params,
		static_cast<float4*>(transforms), context->d_rejected);
	ACLHIP_CHECK_HIP(context, hipGetLastError());
	return ACLHIP_OK;
}
