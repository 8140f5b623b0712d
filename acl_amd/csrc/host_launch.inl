// host_launch.inl -- part of aclhip.hip (one translation unit; included there, in this order, not compiled on its own).
// Host side: pose and single track launches, locality order of instance lists.

namespace
{
	// The shape of a pose launch follows the BATCH, not the registry: a pose row of pose_stride_bytes holds at most pose_stride_bytes /
	// bytes_per_track tracks, so no clip the batch may legally name is larger than that -- and none is larger than the largest registered
	// clip. (The reference sizes its work per clip, decompression.transform.h:1526-1540; until round 4 every launch here was sized for
	// the largest clip of the REGISTRY: one 551-bone crowd leader in the context and every batch of 100-bone characters launched six
	// waves per instance, five of which left after the scalar prologue.) What the shape promises is checked again by the kernels
	// against every clip they meet (launch_refuses_clip, kernels_pose.inl): a clip registered behind a captured launch's back is
	// refused and counted, never decoded into a window or an LDS slot that is too small for it.
	struct pose_launch_shape
	{
		uint32_t windows_per_instance;		// waves per instance: pose windows of the largest pose the batch can hold
		uint32_t lds_quads_per_wave;		// quads of LDS every wave gets for its window's image
		bool wide_key_loads;				// poses of several windows read the bitstream with one aligned request per key (kernels_pose.inl)
		uint32_t items_per_wave;			// poses of several windows, common case kernel: work items a wave takes in turn (1: the one-shot grid)
		bool adjacent_items;				// ... consecutive instances instead of a sweep per turn
	};

	uint32_t batch_pose_quads(const aclhip_context* context, uint32_t layout, uint64_t pose_stride_bytes)
	{
		const uint32_t bytes_per_track = std::max<uint32_t>(aclhip_layout_bytes_per_track(layout), 1);
		const uint64_t stride_quads = pose_stride_bytes / bytes_per_track * 3;
		return uint32_t(std::min<uint64_t>(context->max_pose_quads, stride_quads));
	}

	// (the caller holds the registry lock)
	pose_launch_shape pose_launch_shape_of(const aclhip_context* context, uint32_t layout, uint64_t pose_stride_bytes)
	{
		const uint32_t quads = batch_pose_quads(context, layout, pose_stride_bytes);
		pose_launch_shape shape;
		shape.windows_per_instance = std::max<uint32_t>((quads + k_image_chunk_quads - 1) / k_image_chunk_quads, 1);
		shape.lds_quads_per_wave = std::min<uint32_t>(std::max<uint32_t>(align_to_u32(quads, 64), 64), k_image_chunk_quads);
		// ACLHIP_WIDE_KEY_LOADS=0 / 1 overrides the choice (measurements, parity runs of the other unpack)
		static const int wide_override = []() { const char* value = path_knob("ACLHIP_WIDE_KEY_LOADS"); return value != nullptr ? int(value[0] - '0') : -1; }();
		shape.wide_key_loads = wide_override >= 0 ? wide_override != 0 : shape.windows_per_instance > 1;
		// ACLHIP_IN_TURN_ITEMS = K (0 / 1: one-shot grid), ACLHIP_IN_TURN_ADJACENT = 0 / 1: measurement knobs
		static const uint32_t in_turn_items = []() { const char* value = path_knob("ACLHIP_IN_TURN_ITEMS"); return value != nullptr ? uint32_t(std::atol(value)) : 4u; }();
		static const bool in_turn_adjacent = []() { const char* value = path_knob("ACLHIP_IN_TURN_ADJACENT"); return value != nullptr && value[0] == '1'; }();
		shape.items_per_wave = shape.windows_per_instance > 1 && shape.wide_key_loads ? std::min<uint32_t>(std::max<uint32_t>(in_turn_items, 1), 255) : 1;
		shape.adjacent_items = in_turn_adjacent;
		return shape;
	}

	typedef void (*pose_kernel)(const device_clip*, uint32_t, const uint32_t*, const float*, uint32_t, uint32_t, decode_params, uint8_t*, uint64_t, uint32_t, unsigned long long*);

	// which kernel a pose launch of this shape and these settings takes, and its name (aclhip_describe_tracks_launch: rocprofv3 traces)
	pose_kernel pose_kernel_of(const aclhip_context* context, const decode_params& params, const pose_launch_shape& shape, const char** out_name = nullptr)
	{
		// the common case (track_writer defaults, no per track rounding, normalization != always) copies a resolved pose image
		const bool any_settings = params.standard_defaults == 0 || params.per_track_rounding != 0 || context->force_generic_kernel;
		// an output descriptor that changes nothing (QVV48, nothing skipped) takes the plain kernels
		const bool compact = params.layout != ACLHIP_LAYOUT_QVV48 || params.skip_mask != 0 || params.skip_tracks != nullptr || params.instance_masks != nullptr;
		// the compact layouts with nothing else skipped have kernels that build the LDS image in the output layout (kernels_pose.inl)
		const bool native_layout = !any_settings && params.layout != ACLHIP_LAYOUT_QVV48 && (params.skip_mask & ~(params.layout == ACLHIP_LAYOUT_QV32 ? 4u : 0u)) == 0 && params.skip_tracks == nullptr && params.instance_masks == nullptr;
		const char* name;
		pose_kernel kernel;
		if (native_layout && params.layout == ACLHIP_LAYOUT_QV32) { kernel = decompress_tracks_qv32_kernel; name = "decompress_tracks_qv32_kernel"; }
		else if (native_layout) { kernel = decompress_tracks_qvv40_kernel; name = "decompress_tracks_qvv40_kernel"; }
		else if (any_settings && compact) { kernel = decompress_tracks_any_settings_compact_kernel; name = "decompress_tracks_any_settings_compact_kernel"; }
		else if (any_settings) { kernel = decompress_tracks_any_settings_kernel; name = "decompress_tracks_any_settings_kernel"; }
		else if (compact) { kernel = decompress_tracks_compact_kernel; name = "decompress_tracks_compact_kernel"; }
		// ACLHIP_DECODE_FAST: the plain kernels compiled with the 1 ulp rotation arithmetic (kernels_pose.inl)
		else if (params.fast_math != 0 && shape.wide_key_loads && shape.items_per_wave > 1 && shape.adjacent_items) { kernel = decompress_tracks_in_turn_adjacent_fast_kernel; name = "decompress_tracks_in_turn_adjacent_fast_kernel"; }
		else if (params.fast_math != 0 && shape.wide_key_loads && shape.items_per_wave > 1) { kernel = decompress_tracks_in_turn_fast_kernel; name = "decompress_tracks_in_turn_fast_kernel"; }
		else if (params.fast_math != 0 && shape.wide_key_loads) { kernel = decompress_tracks_wide_loads_fast_kernel; name = "decompress_tracks_wide_loads_fast_kernel"; }
		else if (params.fast_math != 0) { kernel = decompress_tracks_fast_kernel; name = "decompress_tracks_fast_kernel"; }
		else if (shape.wide_key_loads && shape.items_per_wave > 1 && shape.adjacent_items) { kernel = decompress_tracks_in_turn_adjacent_kernel; name = "decompress_tracks_in_turn_adjacent_kernel"; }
		else if (shape.wide_key_loads && shape.items_per_wave > 1) { kernel = decompress_tracks_in_turn_kernel; name = "decompress_tracks_in_turn_kernel"; }
		else if (shape.wide_key_loads) { kernel = decompress_tracks_wide_loads_kernel; name = "decompress_tracks_wide_loads_kernel"; }
		else { kernel = decompress_tracks_kernel; name = "decompress_tracks_kernel"; }
		if (out_name != nullptr)
			*out_name = name;
		return kernel;
	}

	aclhip_status launch_tracks(aclhip_context* context, const aclhip_clip* clips, const float* sample_times, uint32_t num_instances,
		const decode_params& params, void* poses, uint64_t pose_stride_bytes, hipStream_t stream)
	{
		// launches read the registry's maxima and are noted (note_launch_stream) under the registry lock; the clip table itself never moves
		std::shared_lock<std::shared_mutex> lock(context->mutex);
		note_launch_stream(context, stream);

		// one wave per (instance, pose window)
		const pose_launch_shape shape = pose_launch_shape_of(context, params.layout, pose_stride_bytes);
		const uint32_t windows_per_instance = shape.windows_per_instance;
		const uint64_t num_waves = uint64_t(num_instances) * windows_per_instance;
		if (num_waves > 0xFFFFFFFFull - k_waves_per_block)
			return fail(context, ACLHIP_ERROR_INVALID_ARGUMENT, "batch too large: %u instances x %u pose windows", num_instances, windows_per_instance);
		const uint32_t num_blocks = uint32_t((num_waves + k_waves_per_block - 1) / k_waves_per_block);

		const uint32_t lds_quads_per_wave = shape.lds_quads_per_wave;
		size_t lds_bytes = size_t(lds_quads_per_wave) * 16 * k_waves_per_block;
		{
			// measurement aid: ACLHIP_EXTRA_LDS_BYTES inflates a workgroup's LDS so that fewer workgroups fit a CU (occupancy experiments)
			static const size_t extra_lds = []() { const char* value = lab_knob("ACLHIP_EXTRA_LDS_BYTES"); return value != nullptr ? size_t(std::atol(value)) : size_t(0); }();
			lds_bytes += extra_lds;
		}
#if defined(ACLHIP_EXPERIMENTS)
		{
			// round 3's slower kernel variants, selected by environment knobs (tools/experiments/host_experiments.inl)
			const bool any_settings = params.standard_defaults == 0 || params.per_track_rounding != 0 || context->force_generic_kernel;
			const bool compact = params.layout != ACLHIP_LAYOUT_QVV48 || params.skip_mask != 0 || params.skip_tracks != nullptr || params.instance_masks != nullptr;
			bool launched = false;
			const aclhip_status experiment_status = launch_experimental_tracks(context, clips, sample_times, num_instances, params, poses, pose_stride_bytes, stream,
				windows_per_instance, num_waves, num_blocks, any_settings, compact, lds_quads_per_wave, lds_bytes, launched);
			if (launched || experiment_status != ACLHIP_OK)
				return experiment_status;
		}
#endif
		const char* kernel_name = nullptr;
		const pose_kernel kernel = pose_kernel_of(context, params, shape, &kernel_name);
		if (kernel == decompress_tracks_in_turn_kernel || kernel == decompress_tracks_in_turn_adjacent_kernel || kernel == decompress_tracks_in_turn_fast_kernel || kernel == decompress_tracks_in_turn_adjacent_fast_kernel)
		{
			// every wave takes items_per_wave work items in turn (kernels_pose.inl): a K-th of the workgroups
			decode_params turn_params = params;
			turn_params.items_per_wave = uint8_t(shape.items_per_wave);
			uint32_t turn_blocks;
			if (shape.adjacent_items)
			{
				const uint64_t groups = (uint64_t(num_instances) + shape.items_per_wave - 1) / shape.items_per_wave;
				turn_blocks = uint32_t((groups * windows_per_instance + k_waves_per_block - 1) / k_waves_per_block);
			}
			else
			{
				// a wave's items are gridDim * 4 work items apart: a multiple of the windows per instance keeps its window index (and with
				// it the base pose window its LDS image holds) from turn to turn
				// ... and a multiple of the eight XCDs keeps every slot of the list on the XCD the one-shot grid runs it on (what the locality
				// orders of host_launch.inl / kernels_misc.inl are made for)
				turn_blocks = (num_blocks + shape.items_per_wave - 1) / shape.items_per_wave;
				while ((uint64_t(turn_blocks) * k_waves_per_block) % windows_per_instance != 0 || turn_blocks % k_num_xcds != 0)
					turn_blocks++;
			}
			hipLaunchKernelGGL(kernel, dim3(turn_blocks), dim3(k_block_size), lds_bytes, stream,
				context->d_clips, context->d_clips_capacity, clips, sample_times, num_instances, windows_per_instance, turn_params,
				static_cast<uint8_t*>(poses), pose_stride_bytes, lds_quads_per_wave, context->d_rejected);
			ACLHIP_CHECK_HIP(context, hipGetLastError());
			return ACLHIP_OK;
		}
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
		params.skip_tracks = output->skip_tracks;
		// per instance writer decisions (track_writer::skip_track_*(track_index) of ONE pose, core/track_writer.h:189-191; the first K tracks)
		if ((output->instance_masks != nullptr) != (output->mask_table != nullptr))
			return fail(context, ACLHIP_ERROR_INVALID_ARGUMENT, "instance_masks and mask_table come together");
		if (output->instance_masks != nullptr && output->mask_stride == 0)
			return fail(context, ACLHIP_ERROR_INVALID_ARGUMENT, "mask_stride: the bytes of one mask of mask_table (at least the tracks of the largest clip of the batch)");
		if (output->instance_masks != nullptr)
			params.skip_tracks = nullptr;		// (overridden, as the header says)
		params.mask_table = output->mask_table;
		params.instance_masks = output->instance_masks;
		params.mask_stride = output->mask_stride;
		params.instance_track_counts = output->instance_track_counts;
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
// holds k_waves_per_block consecutive (instance, pose window) work items: bucketing the instances by clip and giving every XCD one
// contiguous range of that sequence, served through the slots whose waves start on that XCD, leaves each L2 with an eighth of the
// clips to keep (order_layout, kernels_misc.inl).
namespace
{
	uint32_t windows_per_instance_of(aclhip_context* context)
	{
		if (context == nullptr)
			return 1;
		std::lock_guard<std::shared_mutex> lock(context->mutex);
		return std::max<uint32_t>((context->max_pose_quads + k_image_chunk_quads - 1) / k_image_chunk_quads, 1);
	}

	order_layout make_order_layout(uint32_t num_instances, uint32_t windows_per_instance)
	{
		order_layout layout;
		std::memset(&layout, 0, sizeof(layout));
		uint32_t period = 32;		// 8 workgroups x k_waves_per_block waves, in slots: the smallest p with p * windows a multiple of 32
		for (uint32_t p = 1; p <= 32; ++p)
			if ((uint64_t(p) * windows_per_instance) % (k_num_xcds * k_waves_per_block) == 0)
			{
				period = p;
				break;
			}
		const auto fill = [&](uint32_t slots_per_period, auto&& xcd_of_slot)
		{
			std::memset(layout.per_xcd, 0, sizeof(layout.per_xcd));
			layout.period = slots_per_period;
			for (uint32_t slot = 0; slot < slots_per_period; ++slot)
			{
				const uint32_t xcd = xcd_of_slot(slot);
				layout.slots[xcd][layout.per_xcd[xcd]++] = uint8_t(slot);
			}
		};
		fill(period, [&](uint32_t slot) { return uint32_t((uint64_t(slot) * windows_per_instance / k_waves_per_block) % k_num_xcds); });
		bool every_xcd_starts_poses = true;
		for (uint32_t x = 0; x < k_num_xcds; ++x)
			every_xcd_starts_poses = every_xcd_starts_poses && layout.per_xcd[x] != 0;
		if (!every_xcd_starts_poses)
			fill(k_num_xcds, [](uint32_t slot) { return slot; });		// poses of 8+ windows span the XCDs anyway: plain bucketing by clip

		const uint32_t whole_periods = num_instances / layout.period;
		const uint32_t tail = num_instances % layout.period;
		for (uint32_t x = 0; x < k_num_xcds; ++x)
		{
			uint32_t served = whole_periods * layout.per_xcd[x];
			for (uint32_t k = 0; k < layout.per_xcd[x]; ++k)
				served += layout.slots[x][k] < tail ? 1u : 0u;
			layout.range_begin[x + 1] = layout.range_begin[x] + served;
		}
		return layout;
	}
}

extern "C" aclhip_status aclhip_order_instances_for_locality(const aclhip_context* context, const aclhip_clip* clips, uint32_t num_instances, uint32_t* out_order)
{
	return aclhip_order_instances_for_pose_windows(windows_per_instance_of(const_cast<aclhip_context*>(context)), clips, num_instances, out_order);
}

extern "C" aclhip_status aclhip_order_instances_for_pose_windows(uint32_t windows_per_instance, const aclhip_clip* clips, uint32_t num_instances, uint32_t* out_order)
{
	if ((clips == nullptr || out_order == nullptr) && num_instances != 0)
		return ACLHIP_ERROR_INVALID_ARGUMENT;
	if (windows_per_instance == 0)
		return ACLHIP_ERROR_INVALID_ARGUMENT;
	if (num_instances == 0)
		return ACLHIP_OK;

	const order_layout layout = make_order_layout(num_instances, windows_per_instance);
	return guarded(static_cast<aclhip_context*>(nullptr), [&]() -> aclhip_status
	{
		// the instances bucketed by clip (stable: instances of a clip keep their relative order). A counting sort when the handles
		// are small numbers (they are slots of the registry), a comparison sort for arbitrary values
		uint32_t max_clip = 0;
		for (uint32_t i = 0; i < num_instances; ++i)
			max_clip = std::max(max_clip, clips[i]);
		if (uint64_t(max_clip) < uint64_t(num_instances) * 4 + 1024)
		{
			std::vector<uint32_t> position(size_t(max_clip) + 2, 0);
			for (uint32_t i = 0; i < num_instances; ++i)
				position[size_t(clips[i]) + 1]++;
			for (size_t clip = 0; clip <= max_clip; ++clip)
				position[clip + 1] += position[clip];
			for (uint32_t i = 0; i < num_instances; ++i)
				out_order[order_slot_of(layout, position[clips[i]]++)] = i;
		}
		else
		{
			std::vector<uint32_t> sorted(num_instances);
			for (uint32_t i = 0; i < num_instances; ++i)
				sorted[i] = i;
			std::stable_sort(sorted.begin(), sorted.end(), [&](uint32_t a, uint32_t b) { return clips[a] < clips[b]; });
			for (uint32_t position = 0; position < num_instances; ++position)
				out_order[order_slot_of(layout, position)] = sorted[position];
		}
		return ACLHIP_OK;
	});
}

// Single track requests (decompress_track_kernel): workgroup b takes requests k_block_size b .. k_block_size b + k_block_size - 1 and runs on
// XCD b % 8. The requests bucketed by clip, XCD x serving one contiguous range of that sequence through the workgroups that run on it.
extern "C" aclhip_status aclhip_order_track_requests_for_locality(const aclhip_clip* clips, uint32_t num_requests, uint32_t* out_order)
{
	if ((clips == nullptr || out_order == nullptr) && num_requests != 0)
		return ACLHIP_ERROR_INVALID_ARGUMENT;
	if (num_requests == 0)
		return ACLHIP_OK;
	return guarded(static_cast<aclhip_context*>(nullptr), [&]() -> aclhip_status
	{
		// bucketed by clip, stable (a counting sort when the handles are small numbers -- they are slots of the registry)
		std::vector<uint32_t> sorted(num_requests);
		uint32_t max_clip = 0;
		for (uint32_t i = 0; i < num_requests; ++i)
			max_clip = std::max(max_clip, clips[i]);
		if (uint64_t(max_clip) < uint64_t(num_requests) * 4 + 1024)
		{
			std::vector<uint32_t> position(size_t(max_clip) + 2, 0);
			for (uint32_t i = 0; i < num_requests; ++i)
				position[size_t(clips[i]) + 1]++;
			for (size_t clip = 0; clip <= max_clip; ++clip)
				position[clip + 1] += position[clip];
			for (uint32_t i = 0; i < num_requests; ++i)
				sorted[position[clips[i]]++] = i;
		}
		else
		{
			for (uint32_t i = 0; i < num_requests; ++i)
				sorted[i] = i;
			std::stable_sort(sorted.begin(), sorted.end(), [&](uint32_t a, uint32_t b) { return clips[a] < clips[b]; });
		}

		// how many request slots of the launch run on each XCD, and where each XCD's range of the sorted sequence starts
		const uint64_t whole_blocks = num_requests / k_block_size, tail = num_requests % k_block_size;
		uint64_t cursor[k_num_xcds + 1] = {};
		for (uint32_t x = 0; x < k_num_xcds; ++x)
		{
			uint64_t slots = (whole_blocks / k_num_xcds + (x < whole_blocks % k_num_xcds ? 1 : 0)) * k_block_size;
			if (tail != 0 && whole_blocks % k_num_xcds == x)
				slots += tail;
			cursor[x + 1] = cursor[x] + slots;
		}
		for (uint64_t block = 0; block * k_block_size < num_requests; ++block)
		{
			const uint32_t xcd = uint32_t(block % k_num_xcds);
			const uint64_t first = block * k_block_size, count = std::min<uint64_t>(k_block_size, num_requests - first);
			for (uint64_t k = 0; k < count; ++k)
				out_order[first + k] = sorted[cursor[xcd]++];
		}
		return ACLHIP_OK;
	});
}

// The same order computed on the device, stream ordered: count per clip, scan, scatter (three small kernels on `stream`, counters
// kept per stream by the context). Which instance of a clip takes which of the clip's slots is decided by atomics: a valid order,
// not a reproducible one.
namespace
{
	// windows_per_instance: waves per pose of the launch the order is for (0: the largest registered clip's)
	aclhip_status order_instances_on_device(aclhip_context* context, uint32_t windows_per_instance, const aclhip_clip* clips, const float* sample_times, uint32_t num_instances,
		uint32_t* out_order, aclhip_clip* out_clips, float* out_sample_times, uint32_t* out_positions, void* stream_handle);
}

extern "C" aclhip_status aclhip_order_instances_device(aclhip_context* context, const aclhip_clip* clips, const float* sample_times, uint32_t num_instances,
	uint32_t* out_order, aclhip_clip* out_clips, float* out_sample_times, void* stream_handle)
{
	return order_instances_on_device(context, 0, clips, sample_times, num_instances, out_order, out_clips, out_sample_times, nullptr, stream_handle);
}

extern "C" aclhip_status aclhip_order_instances_device_for_windows(aclhip_context* context, uint32_t windows_per_instance, const aclhip_clip* clips, const float* sample_times,
	uint32_t num_instances, uint32_t* out_order, aclhip_clip* out_clips, float* out_sample_times, void* stream_handle)
{
	if (context == nullptr)
		return ACLHIP_ERROR_INVALID_ARGUMENT;
	if (windows_per_instance == 0)
		return fail(context, ACLHIP_ERROR_INVALID_ARGUMENT, "a pose takes at least one window");
	return order_instances_on_device(context, windows_per_instance, clips, sample_times, num_instances, out_order, out_clips, out_sample_times, nullptr, stream_handle);
}

extern "C" aclhip_status aclhip_pose_windows_of_launch(aclhip_context* context, uint32_t layout, uint64_t pose_stride_bytes, uint32_t* out_windows_per_instance)
{
	if (context == nullptr || out_windows_per_instance == nullptr)
		return ACLHIP_ERROR_INVALID_ARGUMENT;
	if (aclhip_layout_bytes_per_track(layout) == 0)
		return fail(context, ACLHIP_ERROR_INVALID_ARGUMENT, "unknown pose layout %u", layout);
	std::lock_guard<std::shared_mutex> lock(context->mutex);
	*out_windows_per_instance = pose_launch_shape_of(context, layout, pose_stride_bytes).windows_per_instance;
	return ACLHIP_OK;
}

namespace
{
aclhip_status order_instances_on_device(aclhip_context* context, uint32_t windows_per_instance, const aclhip_clip* clips, const float* sample_times, uint32_t num_instances,
	uint32_t* out_order, aclhip_clip* out_clips, float* out_sample_times, uint32_t* out_positions, void* stream_handle)
{
	if (context == nullptr)
		return ACLHIP_ERROR_INVALID_ARGUMENT;
	if (num_instances == 0)
		return ACLHIP_OK;
	if (clips == nullptr || out_order == nullptr || (out_sample_times != nullptr && sample_times == nullptr))
		return fail(context, ACLHIP_ERROR_INVALID_ARGUMENT, "null instance list or order buffer");

	hipStream_t stream = static_cast<hipStream_t>(stream_handle);
	const order_layout layout = make_order_layout(num_instances, windows_per_instance != 0 ? windows_per_instance : windows_per_instance_of(context));

	device_guard guard(context->device);
	std::lock_guard<std::shared_mutex> lock(context->mutex);		// the scratch of a stream is handed to one call at a time, in stream order
	const uint32_t num_bins = uint32_t(context->clips.size()) + 1;		// handles are slots of the registry; the last bin takes everything else
	note_launch_stream(context, stream);
	aclhip_context::order_scratch* scratch = nullptr;
	for (aclhip_context::order_scratch& known : context->order_scratches)
		if (known.stream == stream)
			scratch = &known;
	if (scratch == nullptr)
	{
		context->order_scratches.emplace_back();
		scratch = &context->order_scratches.back();
		scratch->stream = stream;
	}
	const auto reserve = [&](size_t words) -> aclhip_status
	{
		if (scratch->capacity >= words)
			return ACLHIP_OK;
		if (scratch->bins != nullptr)
		{
			aclhip_context::retired_item item;
			item.device_memory = scratch->bins;
			retire(context, std::move(item));
			scratch->bins = nullptr;
			scratch->capacity = 0;
		}
		ACLHIP_CHECK_HIP(context, hipMalloc(reinterpret_cast<void**>(&scratch->bins), words * sizeof(uint32_t)));
		scratch->capacity = words;
		scratch->zeroed_bins = 0;
		return ACLHIP_OK;
	};

	// ACLHIP_ORDER_LAUNCHES=3 forces the older form (three launches: LDS hash tables + device scope atomics), what clip tables of more
	// than k_order_direct_bins entries take anyway
	static const int forced_form = []() { const char* value = path_knob("ACLHIP_ORDER_LAUNCHES"); return value != nullptr ? int(value[0] - '0') : 0; }();
	// A kernel of the one launch form gave up at a barrier since the last call on this stream (its workgroups did not all become
	// resident within seconds: order_grid_barrier): the order it was to write is not there. Said loudly, once; the barrier words are
	// reset and this stream orders with the three launch form -- whose workgroups never wait for one another -- from now on.
	if (scratch->host_failed != nullptr && __atomic_load_n(scratch->host_failed, __ATOMIC_RELAXED) != 0)
	{
		__atomic_store_n(scratch->host_failed, 0u, __ATOMIC_RELAXED);
		scratch->one_launch_form_disabled = true;
		if (scratch->barrier != nullptr)
			(void)hipMemsetAsync(scratch->barrier, 0, sizeof(order_control), stream);
		return fail(context, ACLHIP_ERROR_DEVICE, "an earlier aclhip_order_instances_device on this stream did not complete (its workgroups could not all become resident "
			"within seconds, or -- three launch form -- another ordering wrote this stream's counters at the same time: a captured ordering replayed next to one of "
			"the stream whose scratch it holds): the order it was to write is invalid. This stream orders with three launches from now on; order again");
	}
	if (num_bins <= k_order_direct_bins && forced_form != 3 && !scratch->one_launch_form_disabled)
	{
		// as many workgroups as keep the matrix small (every workgroup reads all of it), all of them resident at once: never more than
		// an idle device holds together (occupancy query: one workgroup of 1 024 threads and 33 KB of LDS per CU at least)
		static const uint32_t max_log2_blocks = []() { const char* value = lab_knob("ACLHIP_ORDER_GRID_LOG2_BLOCKS"); return value != nullptr ? uint32_t(std::atol(value)) : k_order_grid_max_log2_blocks; }();
		static const int blocks_per_cu = []()
		{
			int blocks = 0;
			if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks, reinterpret_cast<const void*>(order_instances_grid_kernel), int(k_order_direct_block_size), 0) != hipSuccess)
			{
				(void)hipGetLastError();
				blocks = 0;
			}
			return blocks;
		}();
		const uint64_t resident_blocks = uint64_t(std::max(blocks_per_cu, 0)) * context->num_compute_units;
		uint32_t log2_blocks = 0;
		while (log2_blocks < max_log2_blocks && (2u << log2_blocks) * k_order_direct_block_size <= num_instances
			&& (size_t(num_bins) << (log2_blocks + 1)) <= k_order_grid_entries && (2u << log2_blocks) <= context->num_compute_units && (2u << log2_blocks) <= resident_blocks)
			++log2_blocks;
		if (resident_blocks != 0)
		{
			const uint32_t num_blocks = 1u << log2_blocks;
			const uint32_t instances_per_block = (num_instances + num_blocks - 1) / num_blocks;
			// the matrix | the bins' totals, at their LARGEST once and for all: 2^19 matrix entries (the loop above) + 8 192 totals, 2 MiB.
			// The scratch of this form then never moves -- a captured hipGraph that holds its address stays valid whatever is registered
			// later -- and no call after the first allocates. (Only registries beyond 8 192 clips, the three launch form, can outgrow it.)
			constexpr size_t k_order_grid_scratch_words = size_t(k_order_grid_entries) + k_order_direct_bins;
			if ((size_t(num_bins) << log2_blocks) + num_bins > k_order_grid_scratch_words)
				return fail(context, ACLHIP_ERROR_DEVICE, "internal: the ordering matrix outgrew its scratch (%u bins x %u workgroups)", num_bins, num_blocks);
			const aclhip_status status = reserve(k_order_grid_scratch_words);
			if (status != ACLHIP_OK)
				return status;
			// (each on its own: a call that fails half way must not leave the next one a barrier without its host word)
			if (scratch->host_failed == nullptr)
			{
				ACLHIP_CHECK_HIP(context, hipHostMalloc(reinterpret_cast<void**>(&scratch->host_failed), sizeof(uint32_t), hipHostMallocMapped));
				*scratch->host_failed = 0;
			}
			if (scratch->barrier == nullptr)
			{
				void* barrier = nullptr;
				ACLHIP_CHECK_HIP(context, hipMalloc(&barrier, sizeof(order_control)));
				hipError_t zeroed = hipMemsetAsync(barrier, 0, sizeof(order_control), context->copy_stream);		// not the caller's stream: it may be capturing
				if (zeroed == hipSuccess)
					zeroed = hipStreamSynchronize(context->copy_stream);
				if (zeroed != hipSuccess)
				{
					(void)hipFree(barrier);
					ACLHIP_CHECK_HIP(context, zeroed);
				}
				scratch->barrier = static_cast<decltype(scratch->barrier)>(barrier);
			}
			scratch->zeroed_bins = 0;		// (the three launch form finds its counters dirty)
			// testing aid: ACLHIP_ORDER_TEST_ABSENT_BLOCK=b keeps workgroup b away from the barriers, ACLHIP_ORDER_TEST_MAX_POLLS shortens the wait
			static const uint32_t absent_block = []() { const char* value = path_knob("ACLHIP_ORDER_TEST_ABSENT_BLOCK"); return value != nullptr ? uint32_t(std::atol(value)) : 0xFFFFFFFFu; }();
			static const uint32_t max_polls = []() { const char* value = path_knob("ACLHIP_ORDER_TEST_MAX_POLLS"); return value != nullptr ? uint32_t(std::atol(value)) : k_order_barrier_max_polls; }();
			hipLaunchKernelGGL(order_instances_grid_kernel, dim3(num_blocks), dim3(k_order_direct_block_size), 0, stream, clips, sample_times, num_instances, instances_per_block,
				num_bins, log2_blocks, scratch->bins, reinterpret_cast<order_control*>(scratch->barrier), scratch->host_failed, max_polls, absent_block, layout, out_order, out_clips, out_sample_times, out_positions);
			ACLHIP_CHECK_HIP(context, hipGetLastError());
			return ACLHIP_OK;
		}
	}
	const size_t padded_bins = (size_t(num_bins) + 4095) / 4096 * 4096;
	{
		const aclhip_status status = reserve(padded_bins * 2);
		if (status != ACLHIP_OK)
			return status;
	}
	if (scratch->zeroed_bins != padded_bins)
	{
		ACLHIP_CHECK_HIP(context, hipMemsetAsync(scratch->bins, 0, (padded_bins * 2) * sizeof(uint32_t), stream));		// once: every call leaves the counters at zero
		scratch->zeroed_bins = padded_bins;
	}
	uint32_t* counters = scratch->bins;
	uint32_t* cursors = scratch->bins + padded_bins;
	// (the three launch form reports a corrupted ordering through the same host word as the one launch form's barrier give-up)
	if (scratch->host_failed == nullptr)
	{
		ACLHIP_CHECK_HIP(context, hipHostMalloc(reinterpret_cast<void**>(&scratch->host_failed), sizeof(uint32_t), hipHostMallocMapped));
		*scratch->host_failed = 0;
	}

	const uint32_t num_blocks = (num_instances + k_order_instances_per_block - 1) / k_order_instances_per_block;
	hipLaunchKernelGGL(order_count_kernel, dim3(num_blocks), dim3(k_order_block_size), 0, stream, clips, num_instances, num_bins, counters);
	hipLaunchKernelGGL(order_scan_kernel, dim3(1), dim3(1024), 0, stream, counters, cursors, num_bins);
	hipLaunchKernelGGL(order_scatter_kernel, dim3(num_blocks), dim3(k_order_block_size), 0, stream, clips, sample_times, num_instances, num_bins, cursors, layout, out_order, out_clips, out_sample_times, out_positions, scratch->host_failed);
	ACLHIP_CHECK_HIP(context, hipGetLastError());
	return ACLHIP_OK;
}
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

	std::shared_lock<std::shared_mutex> lock(context->mutex);		// see launch_tracks
	device_guard guard(context->device);
	note_launch_stream(context, static_cast<hipStream_t>(stream));
	const uint32_t num_blocks = (num_instances + k_block_size - 1) / k_block_size;
	// (ACLHIP_DECODE_FAST changes nothing here: the variant compiled for it never measured faster than the bit exact kernel, kernels_track.inl)
	hipLaunchKernelGGL(decompress_track_kernel, dim3(num_blocks), dim3(k_block_size), 0, static_cast<hipStream_t>(stream),
		context->d_clips, context->d_clips_capacity, clips, sample_times, track_indices, num_instances, device_params,
		static_cast<float4*>(transforms), context->d_rejected);
	ACLHIP_CHECK_HIP(context, hipGetLastError());
	return ACLHIP_OK;
}
