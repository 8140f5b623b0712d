// host_scalar_misc.inl -- part of aclhip.hip (one translation unit; included there, in this order, not compiled on its own).
// Host side: scalar track list launches, all samples, all-gather, measurement helpers.

// ---- scalar track lists --------------------------------------------------------------------------------------------

namespace
{
	aclhip_status launch_scalar(aclhip_context* context, const aclhip_clip* clips, const float* sample_times, const uint32_t* track_indices, uint32_t num_instances,
		const aclhip_decompress_params* params, void* out, uint64_t out_stride_bytes, void* stream)
	{
		if (context == nullptr)
			return ACLHIP_ERROR_INVALID_ARGUMENT;
		if (num_instances == 0)
			return ACLHIP_OK;
		if (clips == nullptr || sample_times == nullptr || out == nullptr)
			return fail(context, ACLHIP_ERROR_INVALID_ARGUMENT, "null instance list or output buffer");
		if ((reinterpret_cast<uintptr_t>(out) & 3u) != 0 || (out_stride_bytes & 3u) != 0 || out_stride_bytes == 0)
			return fail(context, ACLHIP_ERROR_INVALID_ARGUMENT, "the output buffer and its stride must be 4 byte aligned");

		decode_params device_params;
		const aclhip_status status = resolve_params(context, params, device_params);
		if (status != ACLHIP_OK)
			return status;

		std::shared_lock<std::shared_mutex> lock(context->mutex);		// see launch_tracks
		device_guard guard(context->device);
		note_launch_stream(context, static_cast<hipStream_t>(stream));
		if (track_indices != nullptr)
		{
			const uint32_t num_blocks = (num_instances + k_block_size - 1) / k_block_size;
			hipLaunchKernelGGL(decompress_scalar_track_kernel, dim3(num_blocks), dim3(k_block_size), 0, static_cast<hipStream_t>(stream),
				context->d_clips, context->d_clips_capacity, clips, sample_times, track_indices, num_instances, device_params,
				static_cast<uint8_t*>(out), out_stride_bytes, context->d_rejected);
		}
		else
		{
			// one wave per (instance, 64 or 256 tracks). Sized for the BATCH like the pose launches (pose_launch_shape_of): a row of
			// out_stride_bytes holds at most out_stride_bytes / 4 tracks, a frame at most 32 bits per float of a row -- and neither is larger
			// than the largest registered list's; the kernels check every list they meet against it (scalar_launch_refuses_clip)
			const uint32_t batch_tracks = uint32_t(std::min<uint64_t>(context->max_scalar_tracks, out_stride_bytes / 4));
			const uint32_t batch_frame_bytes = uint32_t(std::min<uint64_t>(context->max_scalar_frame_bytes, out_stride_bytes));
			const uint32_t rows = batch_tracks <= k_wave_size ? 1u : k_scalar_tracks_per_wave / k_wave_size;
			const uint32_t tracks_per_wave = rows * k_wave_size;
			const uint32_t chunks_per_instance = std::max<uint32_t>((batch_tracks + tracks_per_wave - 1) / tracks_per_wave, 1);
			const uint64_t num_waves = uint64_t(num_instances) * chunks_per_instance;
			if (num_waves > 0xFFFFFFFFull - k_waves_per_block)
				return fail(context, ACLHIP_ERROR_INVALID_ARGUMENT, "batch too large: %u instances x %u track chunks", num_instances, chunks_per_instance);

			// LDS copy of both key frames per wave: the frame, the 16 byte alignment slack in front, 8 bytes behind. Skipped (global
			// reads) when that would leave fewer than 4 workgroups per CU, and for short frames, where a handful of scattered reads
			// is cheaper than the copy and its barrier (measured: 64 float1f tracks 16 vs 23 us, 256 tracks 38 vs 32 us)
			uint32_t frame_lds_bytes = align_to_u32(batch_frame_bytes + 16 + 8, 16);
			if (size_t(frame_lds_bytes) * 2 * k_waves_per_block > 40 * 1024 || batch_frame_bytes < 192)
				frame_lds_bytes = 0;

			// groups of k_scalar_group consecutive instances per wave (kernels_scalar.inl) when the frames of a group fit LDS with at least
			// four workgroups per CU and the batch is large enough to fill the device with a quarter of the waves
			const bool grouped = frame_lds_bytes != 0 && size_t(frame_lds_bytes) * 2 * k_scalar_group * k_waves_per_block <= 40 * 1024 && num_instances >= 16384;
			const uint64_t launch_waves = grouped ? uint64_t((num_instances + k_scalar_group - 1) / k_scalar_group) * chunks_per_instance : num_waves;
			const dim3 grid(uint32_t((launch_waves + k_waves_per_block - 1) / k_waves_per_block));
			const size_t lds_bytes = size_t(frame_lds_bytes) * 2 * k_waves_per_block * (grouped ? k_scalar_group : 1u);
			const auto launch = [&](auto kernel)
			{
				hipLaunchKernelGGL(kernel, grid, dim3(k_block_size), lds_bytes, static_cast<hipStream_t>(stream), context->d_clips, context->d_clips_capacity, clips, sample_times,
					num_instances, chunks_per_instance, device_params, static_cast<uint8_t*>(out), out_stride_bytes, frame_lds_bytes, context->d_rejected);
			};
			const bool policies = device_params.per_track_rounding != 0;
			if (grouped && context->num_wide_scalar_clips == 0)
			{
				// every registered list is float1f (blend shape weights, curves): the kernel compiled for one float per track only
				// (47 instead of 109 registers: twice the waves per SIMD)
				if (rows == 1) { if (policies) launch(decompress_scalar_tracks_grouped_kernel<1, true, 1>); else launch(decompress_scalar_tracks_grouped_kernel<1, false, 1>); }
				else { if (policies) launch(decompress_scalar_tracks_grouped_kernel<4, true, 1>); else launch(decompress_scalar_tracks_grouped_kernel<4, false, 1>); }
			}
			else if (grouped)
			{
				if (rows == 1) { if (policies) launch(decompress_scalar_tracks_grouped_kernel<1, true, 0>); else launch(decompress_scalar_tracks_grouped_kernel<1, false, 0>); }
				else { if (policies) launch(decompress_scalar_tracks_grouped_kernel<4, true, 0>); else launch(decompress_scalar_tracks_grouped_kernel<4, false, 0>); }
			}
			else if (frame_lds_bytes != 0)
			{
				if (rows == 1) { if (policies) launch(decompress_scalar_tracks_kernel<true, 1, true>); else launch(decompress_scalar_tracks_kernel<true, 1, false>); }
				else { if (policies) launch(decompress_scalar_tracks_kernel<true, 4, true>); else launch(decompress_scalar_tracks_kernel<true, 4, false>); }
			}
			else
			{
				if (rows == 1) { if (policies) launch(decompress_scalar_tracks_kernel<false, 1, true>); else launch(decompress_scalar_tracks_kernel<false, 1, false>); }
				else { if (policies) launch(decompress_scalar_tracks_kernel<false, 4, true>); else launch(decompress_scalar_tracks_kernel<false, 4, false>); }
			}
		}
		ACLHIP_CHECK_HIP(context, hipGetLastError());
		return ACLHIP_OK;
	}

	// Host pointer convenience for scalar track lists: uploads the instance lists and the caller's buffer (values the decode does
	// not write keep what the caller had), runs the batch, downloads.
	aclhip_status decompress_scalar_host(aclhip_context* context, const aclhip_clip* clips, const float* sample_times, const uint32_t* track_indices, uint32_t num_instances,
		const aclhip_decompress_params* params, void* out, uint64_t out_stride_bytes)
	{
		if (context == nullptr)
			return ACLHIP_ERROR_INVALID_ARGUMENT;
		if (num_instances == 0)
			return ACLHIP_OK;
		if (clips == nullptr || sample_times == nullptr || out == nullptr || out_stride_bytes == 0)
			return fail(context, ACLHIP_ERROR_INVALID_ARGUMENT, "null instance list or output buffer");

		aclhip_decompress_params local;
		if (params != nullptr) local = *params; else aclhip_default_params(&local);
		local.default_values = nullptr;

		uint32_t max_tracks = 0;
		{
			std::lock_guard<std::shared_mutex> lock(context->mutex);
			for (uint32_t i = 0; i < num_instances; ++i)
				if (clips[i] < context->clips.size() && context->clips[clips[i]].in_use)
					max_tracks = std::max(max_tracks, context->clips[clips[i]].info.num_tracks);
		}

		device_guard guard(context->device);
		// (the host convenience calls are synchronous by contract and stage through temporary device buffers on the default stream;
		// callers that must not disturb work in flight use the device pointer entry points)
		hipStream_t work_stream = nullptr;
		std::vector<void*> allocations;
		auto release = [&]() { for (void* p : allocations) (void)hipFree(p); };
		auto upload = [&](const void* host, size_t bytes, void** out_device) -> bool
		{
			void* d = nullptr;
			if (hipMalloc(&d, std::max<size_t>(bytes, 16)) != hipSuccess)
				return false;
			allocations.push_back(d);
			if (host != nullptr && hipMemcpy(d, host, bytes, hipMemcpyHostToDevice) != hipSuccess)
				return false;
			*out_device = d;
			return true;
		};

		const size_t out_bytes = size_t(out_stride_bytes) * num_instances;
		void* d_clip_ids = nullptr; void* d_times = nullptr; void* d_tracks = nullptr; void* d_out = nullptr;
		void* d_track_policies = nullptr; void* d_instance_policies = nullptr;
		bool ok = upload(clips, sizeof(uint32_t) * num_instances, &d_clip_ids) && upload(sample_times, sizeof(float) * num_instances, &d_times)
			&& upload(out, out_bytes, &d_out);
		if (ok && track_indices != nullptr)
			ok = upload(track_indices, sizeof(uint32_t) * num_instances, &d_tracks);
		if (ok && local.track_rounding_policies != nullptr)
			ok = upload(local.track_rounding_policies, std::max<uint32_t>(max_tracks, 1), &d_track_policies);
		if (ok && local.instance_rounding_policies != nullptr)
			ok = upload(local.instance_rounding_policies, num_instances, &d_instance_policies);
		void* d_instance_looping = nullptr;
		if (ok && local.instance_looping_policies != nullptr)
			ok = upload(local.instance_looping_policies, num_instances, &d_instance_looping);
		// per instance writers' track rounding tables: as many tables as the largest index names
		void* d_rounding_table = nullptr; void* d_rounding_tables_of = nullptr;
		if (ok && local.track_rounding_table != nullptr && local.instance_rounding_tables != nullptr)
		{
			uint32_t num_tables = 0;
			for (uint32_t i = 0; i < num_instances; ++i)
				num_tables = std::max<uint32_t>(num_tables, uint32_t(local.instance_rounding_tables[i]) + 1);
			ok = upload(local.instance_rounding_tables, num_instances, &d_rounding_tables_of) && upload(local.track_rounding_table, size_t(num_tables) * local.track_rounding_stride, &d_rounding_table);
		}

		if (!ok)
		{
			release();
			return fail(context, ACLHIP_ERROR_DEVICE, "staging the batch on the device failed");
		}
		local.track_rounding_policies = static_cast<const uint8_t*>(d_track_policies);
		local.instance_rounding_policies = static_cast<const uint8_t*>(d_instance_policies);
		local.instance_looping_policies = static_cast<const uint8_t*>(d_instance_looping);
		if (local.track_rounding_table != nullptr && local.instance_rounding_tables != nullptr)
		{
			local.track_rounding_table = static_cast<const uint8_t*>(d_rounding_table);
			local.instance_rounding_tables = static_cast<const uint8_t*>(d_rounding_tables_of);
		}

		aclhip_status status = launch_scalar(context, static_cast<const aclhip_clip*>(d_clip_ids), static_cast<const float*>(d_times), static_cast<const uint32_t*>(d_tracks),
			num_instances, &local, d_out, out_stride_bytes, work_stream);
		if (status == ACLHIP_OK)
		{
			hipError_t copy_status = hipStreamSynchronize(work_stream);
			if (copy_status == hipSuccess)
				copy_status = hipMemcpy(out, d_out, out_bytes, hipMemcpyDeviceToHost);
			if (copy_status != hipSuccess)
				status = fail(context, ACLHIP_ERROR_DEVICE, "downloading the values failed: %s", hipGetErrorString(copy_status));
		}
		release();
		return status;
	}
}

extern "C" aclhip_status aclhip_decompress_scalar_tracks_batch(aclhip_context* context, const aclhip_clip* clips, const float* sample_times, uint32_t num_instances,
	const aclhip_decompress_params* params, void* values, uint64_t stride_bytes, void* stream)
{
	return launch_scalar(context, clips, sample_times, nullptr, num_instances, params, values, stride_bytes, stream);
}

extern "C" aclhip_status aclhip_decompress_scalar_track_batch(aclhip_context* context, const aclhip_clip* clips, const float* sample_times, const uint32_t* track_indices,
	uint32_t num_instances, const aclhip_decompress_params* params, void* values, uint64_t stride_bytes, void* stream)
{
	if (track_indices == nullptr && num_instances != 0)
		return context != nullptr ? fail(context, ACLHIP_ERROR_INVALID_ARGUMENT, "null track index list") : ACLHIP_ERROR_INVALID_ARGUMENT;
	return launch_scalar(context, clips, sample_times, track_indices, num_instances, params, values, stride_bytes, stream);
}

extern "C" aclhip_status aclhip_decompress_scalar_tracks_host(aclhip_context* context, const aclhip_clip* clips, const float* sample_times, uint32_t num_instances,
	const aclhip_decompress_params* params, void* values, uint64_t stride_bytes)
{
	return decompress_scalar_host(context, clips, sample_times, nullptr, num_instances, params, values, stride_bytes);
}

extern "C" aclhip_status aclhip_decompress_scalar_track_host(aclhip_context* context, const aclhip_clip* clips, const float* sample_times, const uint32_t* track_indices,
	uint32_t num_instances, const aclhip_decompress_params* params, void* values, uint64_t stride_bytes)
{
	if (track_indices == nullptr && num_instances != 0)
		return context != nullptr ? fail(context, ACLHIP_ERROR_INVALID_ARGUMENT, "null track index list") : ACLHIP_ERROR_INVALID_ARGUMENT;
	return decompress_scalar_host(context, clips, sample_times, track_indices, num_instances, params, values, stride_bytes);
}

// ---- every sample of a clip -------------------------------------------------------------------------------------------

extern "C" aclhip_status aclhip_decompress_all_samples(aclhip_context* context, aclhip_clip clip, const aclhip_decompress_params* params,
	void* scratch, void* out, uint64_t stride_bytes, void* stream)
{
	if (context == nullptr)
		return ACLHIP_ERROR_INVALID_ARGUMENT;
	aclhip_clip_info info;
	{
		std::lock_guard<std::shared_mutex> lock(context->mutex);
		if (clip >= context->clips.size() || !context->clips[clip].in_use)
			return fail(context, ACLHIP_ERROR_UNKNOWN_CLIP, "unknown clip handle %u", clip);
		info = context->clips[clip].info;
	}
	if (info.num_tracks == 0 || info.num_samples == 0)
		return ACLHIP_OK;
	if (scratch == nullptr || out == nullptr || (reinterpret_cast<uintptr_t>(scratch) & 3u) != 0)
		return fail(context, ACLHIP_ERROR_INVALID_ARGUMENT, "null or misaligned scratch / output buffer");

	aclhip_decompress_params local;
	if (params != nullptr) local = *params; else aclhip_default_params(&local);
	local.rounding_policy = ACLHIP_ROUND_NEAREST;		// convert.impl.h:166
	local.instance_rounding_policies = nullptr;
	local.instance_looping_policies = nullptr;
	local.track_rounding_table = nullptr;
	local.instance_rounding_tables = nullptr;

	// the duration the reference's loop clamps to is the one of the looping policy in effect (convert.impl.h:139)
	float duration = info.duration;
	if (local.looping_policy != ACLHIP_LOOP_AS_COMPRESSED)
	{
		const uint32_t samples = info.num_samples + (local.looping_policy == ACLHIP_LOOP_WRAP ? 1u : 0u);
		duration = samples <= 1 ? 0.0f : float(samples - 1) / info.sample_rate;
	}

	uint32_t* clip_ids = static_cast<uint32_t*>(scratch);
	float* sample_times = reinterpret_cast<float*>(clip_ids + info.num_samples);
	{
		device_guard guard(context->device);
		hipLaunchKernelGGL(fill_sample_instances_kernel, dim3((info.num_samples + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream),
			clip, info.num_samples, info.sample_rate, duration, clip_ids, sample_times);
		ACLHIP_CHECK_HIP(context, hipGetLastError());
	}

	if (info.track_type == k_track_type_qvvf)
		return aclhip_decompress_tracks_batch(context, clip_ids, sample_times, info.num_samples, &local, out, stride_bytes, stream);
	return aclhip_decompress_scalar_tracks_batch(context, clip_ids, sample_times, info.num_samples, &local, out, stride_bytes, stream);
}

// ---- multi-GPU gather ------------------------------------------------------------------------------------------------

namespace
{
	// RCCL is only ever touched by callers that gather; decoding never loads it. The communicator a caller hands over was made by the
	// RCCL of ITS process, so that is the library whose ncclAllGather must run: a symbol that is already visible wins, then a library
	// that is already loaded under RCCL's soname (RTLD_NOLOAD: PyTorch loads its bundled librccl privately), and only then a fresh load.
	struct rccl_entry_points
	{
		void* library = nullptr;
		int (*all_gather)(const void*, void*, size_t, int, void*, hipStream_t) = nullptr;		// ncclAllGather (rccl.h:678)
		int (*get_version)(int*) = nullptr;															// ncclGetVersion
		const char* how = "";
	};

	const rccl_entry_points& rccl()
	{
		static const rccl_entry_points entry_points = []()
		{
			rccl_entry_points found;
			void* all_gather = dlsym(RTLD_DEFAULT, "ncclAllGather");
			void* get_version = dlsym(RTLD_DEFAULT, "ncclGetVersion");
			found.how = "already visible in the process";
			if (all_gather == nullptr)
			{
				static const char* const names[] = { "librccl.so.1", "librccl.so" };
				for (int pass = 0; pass < 2 && found.library == nullptr; ++pass)
					for (const char* name : names)
					{
						found.library = dlopen(name, RTLD_NOW | RTLD_GLOBAL | (pass == 0 ? RTLD_NOLOAD : 0));
						if (found.library != nullptr)
						{
							found.how = pass == 0 ? "already loaded (found by its soname)" : "loaded by this library";
							break;
						}
					}
				if (found.library != nullptr)
				{
					all_gather = dlsym(found.library, "ncclAllGather");
					get_version = dlsym(found.library, "ncclGetVersion");
				}
			}
			found.all_gather = reinterpret_cast<decltype(found.all_gather)>(all_gather);
			found.get_version = reinterpret_cast<decltype(found.get_version)>(get_version);
			return found;
		}();
		return entry_points;
	}
}

// What aclhip_all_gather_poses would call, without a communicator: resolves ncclAllGather and ncclGetVersion the same way, calls the
// latter. For a check BEFORE a multi-GPU job is leased (tests/test_gpu_all_gather.py): a missing soname or symbol shows here.
// out_version: RCCL's version code (e.g. 22606); out_path: the file the symbol lives in (dladdr), out_how: how it was found.
extern "C" aclhip_status aclhip_probe_rccl(int* out_version, char* out_path, uint32_t path_capacity, char* out_how, uint32_t how_capacity)
{
	const rccl_entry_points& entry_points = rccl();
	if (out_version != nullptr)
		*out_version = 0;
	if (out_path != nullptr && path_capacity != 0)
		out_path[0] = 0;
	if (out_how != nullptr && how_capacity != 0)
		std::snprintf(out_how, how_capacity, "%s", entry_points.how);
	if (entry_points.all_gather == nullptr || entry_points.get_version == nullptr)
		return fail(static_cast<aclhip_context*>(nullptr), ACLHIP_ERROR_DEVICE, "RCCL is not available: %s", entry_points.library == nullptr ? "librccl.so.1 / librccl.so could not be loaded" : "the library has no ncclAllGather / ncclGetVersion");
	int version = 0;
	const int result = entry_points.get_version(&version);
	if (result != 0)
		return fail(static_cast<aclhip_context*>(nullptr), ACLHIP_ERROR_DEVICE, "ncclGetVersion failed: ncclResult_t %d", result);
	if (out_version != nullptr)
		*out_version = version;
	Dl_info info;
	if (out_path != nullptr && path_capacity != 0 && dladdr(reinterpret_cast<void*>(entry_points.all_gather), &info) != 0 && info.dli_fname != nullptr)
		std::snprintf(out_path, path_capacity, "%s", info.dli_fname);
	return ACLHIP_OK;
}

extern "C" aclhip_status aclhip_all_gather_poses(aclhip_context* context, void* rccl_comm, const void* shard_poses, void* all_poses, uint64_t shard_bytes, void* stream)
{
	if (context == nullptr)
		return ACLHIP_ERROR_INVALID_ARGUMENT;
	if (rccl_comm == nullptr || shard_poses == nullptr || all_poses == nullptr)
		return fail(context, ACLHIP_ERROR_INVALID_ARGUMENT, "null communicator or buffer");
	if (shard_bytes == 0)
		return ACLHIP_OK;

	const rccl_entry_points& entry_points = rccl();
	if (entry_points.all_gather == nullptr)
		return fail(context, ACLHIP_ERROR_DEVICE, "RCCL is not available: %s", entry_points.library == nullptr ? "librccl.so.1 / librccl.so could not be loaded" : "the library has no ncclAllGather");

	device_guard guard(context->device);
	constexpr int k_nccl_uint8 = 1;		// ncclUint8 (rccl.h:460)
	const int result = entry_points.all_gather(shard_poses, all_poses, size_t(shard_bytes), k_nccl_uint8, rccl_comm, static_cast<hipStream_t>(stream));
	if (result != 0)
		return fail(context, ACLHIP_ERROR_DEVICE, "ncclAllGather failed: ncclResult_t %d", result);
	return ACLHIP_OK;
}

// Peer gather: the poses are wanted on ONE GPU (the renderer's). Every other GPU then pushes its shard over its own xGMI link straight
// into that GPU's buffer -- seven links into the destination work concurrently, nothing travels twice -- instead of a ring collective
// that also hands every shard to every rank. One process per GPU: the destination exports its buffer as a HIP IPC handle
// (64 bytes, carried by whatever the processes already talk over), the others map it and copy device to device.
extern "C" aclhip_status aclhip_peer_export_buffer(aclhip_context* context, void* device_buffer, uint8_t* out_handle)
{
	if (context == nullptr)
		return ACLHIP_ERROR_INVALID_ARGUMENT;
	if (device_buffer == nullptr || out_handle == nullptr)
		return fail(context, ACLHIP_ERROR_INVALID_ARGUMENT, "null buffer or handle");
	static_assert(sizeof(hipIpcMemHandle_t) + sizeof(uint64_t) == ACLHIP_PEER_HANDLE_BYTES, "handle size");
	device_guard guard(context->device);
	// an IPC handle names a whole allocation: buffers carved out of a larger one (a caching allocator's block) travel as base + offset
	hipDeviceptr_t base = nullptr;
	size_t allocation_size = 0;
	ACLHIP_CHECK_HIP(context, hipMemGetAddressRange(&base, &allocation_size, device_buffer));
	hipIpcMemHandle_t handle;
	ACLHIP_CHECK_HIP(context, hipIpcGetMemHandle(&handle, base));
	const uint64_t offset = uint64_t(static_cast<const uint8_t*>(device_buffer) - static_cast<const uint8_t*>(base));
	std::memcpy(out_handle, &handle, sizeof(handle));
	std::memcpy(out_handle + sizeof(handle), &offset, sizeof(offset));
	return ACLHIP_OK;
}

extern "C" aclhip_status aclhip_peer_open_buffer(aclhip_context* context, const uint8_t* handle, void** out_device_buffer)
{
	if (context == nullptr)
		return ACLHIP_ERROR_INVALID_ARGUMENT;
	if (handle == nullptr || out_device_buffer == nullptr)
		return fail(context, ACLHIP_ERROR_INVALID_ARGUMENT, "null handle");
	*out_device_buffer = nullptr;
	device_guard guard(context->device);
	hipIpcMemHandle_t ipc_handle;
	uint64_t offset = 0;
	std::memcpy(&ipc_handle, handle, sizeof(ipc_handle));
	std::memcpy(&offset, handle + sizeof(ipc_handle), sizeof(offset));
	void* base = nullptr;
	ACLHIP_CHECK_HIP(context, hipIpcOpenMemHandle(&base, ipc_handle, hipIpcMemLazyEnablePeerAccess));
	std::lock_guard<std::shared_mutex> lock(context->mutex);
	context->peer_mappings.push_back({ static_cast<uint8_t*>(base) + offset, base });
	*out_device_buffer = static_cast<uint8_t*>(base) + offset;
	return ACLHIP_OK;
}

extern "C" aclhip_status aclhip_peer_close_buffer(aclhip_context* context, void* device_buffer)
{
	if (context == nullptr)
		return ACLHIP_ERROR_INVALID_ARGUMENT;
	if (device_buffer == nullptr)
		return ACLHIP_OK;
	void* base = nullptr;
	{
		std::lock_guard<std::shared_mutex> lock(context->mutex);
		for (size_t i = 0; i < context->peer_mappings.size(); ++i)
			if (context->peer_mappings[i].buffer == device_buffer)
			{
				base = context->peer_mappings[i].base;
				context->peer_mappings.erase(context->peer_mappings.begin() + ptrdiff_t(i));
				break;
			}
	}
	if (base == nullptr)
		return fail(context, ACLHIP_ERROR_INVALID_ARGUMENT, "not a buffer aclhip_peer_open_buffer returned");
	device_guard guard(context->device);
	ACLHIP_CHECK_HIP(context, hipIpcCloseMemHandle(base));
	return ACLHIP_OK;
}

extern "C" aclhip_status aclhip_push_poses_to_peer(aclhip_context* context, void* peer_buffer, uint64_t offset_bytes, const void* shard_poses, uint64_t shard_bytes, void* stream)
{
	if (context == nullptr)
		return ACLHIP_ERROR_INVALID_ARGUMENT;
	if (peer_buffer == nullptr || shard_poses == nullptr)
		return fail(context, ACLHIP_ERROR_INVALID_ARGUMENT, "null buffer");
	if (shard_bytes == 0)
		return ACLHIP_OK;
	device_guard guard(context->device);
	// unified addressing: the runtime sees a peer mapped destination and drives the copy over the link between the two GPUs (SDMA)
	ACLHIP_CHECK_HIP(context, hipMemcpyAsync(static_cast<uint8_t*>(peer_buffer) + offset_bytes, shard_poses, shard_bytes, hipMemcpyDeviceToDevice, static_cast<hipStream_t>(stream)));
	return ACLHIP_OK;
}

extern "C" aclhip_status aclhip_forget_stream(aclhip_context* context, void* stream)
{
	if (context == nullptr)
		return ACLHIP_ERROR_INVALID_ARGUMENT;
	hipStream_t hip_stream = static_cast<hipStream_t>(stream);
	device_guard guard(context->device);
	// (outside the lock: other threads keep registering and launching while this stream drains)
	ACLHIP_CHECK_HIP(context, hipStreamSynchronize(hip_stream));
	std::lock_guard<std::shared_mutex> lock(context->mutex);
	for (size_t i = 0; i < context->launch_streams.size(); ++i)
		if (context->launch_streams[i] == hip_stream)
		{
			context->launch_streams.erase(context->launch_streams.begin() + ptrdiff_t(i));
			break;
		}
	for (size_t i = 0; i < context->order_scratches.size(); ++i)
		if (context->order_scratches[i].stream == hip_stream)
		{
			if (context->order_scratches[i].bins != nullptr)
				(void)hipFree(context->order_scratches[i].bins);		// the stream is idle: nothing uses its scratch
			if (context->order_scratches[i].barrier != nullptr)
				(void)hipFree(context->order_scratches[i].barrier);
			if (context->order_scratches[i].host_failed != nullptr)
				(void)hipHostFree(context->order_scratches[i].host_failed);
			context->order_scratches.erase(context->order_scratches.begin() + ptrdiff_t(i));
			break;
		}
	collect_retired(context, false);
	return ACLHIP_OK;
}

extern "C" aclhip_status aclhip_get_lifetime_stats(aclhip_context* context, uint64_t* out_stats)
{
	if (context == nullptr || out_stats == nullptr)
		return ACLHIP_ERROR_INVALID_ARGUMENT;
	std::lock_guard<std::shared_mutex> lock(context->mutex);
	device_guard guard(context->device);
	collect_retired(context, false);
	out_stats[0] = context->clips_registered;
	out_stats[1] = context->clips_unregistered;
	out_stats[2] = context->deferred_frees_completed;
	out_stats[3] = context->retired.size();
	out_stats[4] = context->d_clips_capacity;
	out_stats[5] = context->table_is_virtual ? 1 : 0;
	out_stats[6] = reinterpret_cast<uintptr_t>(context->d_clips);
	out_stats[7] = context->launch_streams.size();
	return ACLHIP_OK;
}

extern "C" aclhip_status aclhip_get_rejected_instance_count(aclhip_context* context, uint64_t* out_count)
{
	if (context == nullptr || out_count == nullptr)
		return ACLHIP_ERROR_INVALID_ARGUMENT;
	device_guard guard(context->device);
	unsigned long long value = 0;
	ACLHIP_CHECK_HIP(context, hipDeviceSynchronize());
	ACLHIP_CHECK_HIP(context, hipMemcpy(&value, context->d_rejected, sizeof(value), hipMemcpyDeviceToHost));
	*out_count = value;
	return ACLHIP_OK;
}

extern "C" aclhip_status aclhip_get_negative_scale_count(aclhip_context* context, uint64_t* out_count)
{
	if (context == nullptr || out_count == nullptr)
		return ACLHIP_ERROR_INVALID_ARGUMENT;
	device_guard guard(context->device);
	unsigned long long value = 0;
	ACLHIP_CHECK_HIP(context, hipDeviceSynchronize());
	ACLHIP_CHECK_HIP(context, hipMemcpy(&value, context->d_rejected + 1, sizeof(value), hipMemcpyDeviceToHost));
	*out_count = value;
	return ACLHIP_OK;
}

extern "C" aclhip_status aclhip_time_decompress_tracks_batch(aclhip_context* context, const aclhip_clip* clips, const float* sample_times, uint32_t num_instances,
	const aclhip_decompress_params* params, void* poses, uint64_t pose_stride_bytes, void* stream, uint32_t repeats, float* out_ms_per_launch)
{
	if (out_ms_per_launch == nullptr || repeats == 0)
		return ACLHIP_ERROR_INVALID_ARGUMENT;
	aclhip_status status = check_batch_arguments(context, clips, sample_times, num_instances, poses, pose_stride_bytes);
	if (status != ACLHIP_OK)
		return status;

	decode_params device_params;
	status = resolve_params(context, params, device_params);
	if (status != ACLHIP_OK)
		return status;

	device_guard guard(context->device);
	hipStream_t hip_stream = static_cast<hipStream_t>(stream);
	hipEvent_t start, stop;
	ACLHIP_CHECK_HIP(context, hipEventCreate(&start));
	ACLHIP_CHECK_HIP(context, hipEventCreate(&stop));
	ACLHIP_CHECK_HIP(context, hipEventRecord(start, hip_stream));
	for (uint32_t i = 0; i < repeats && status == ACLHIP_OK; ++i)
		status = launch_tracks(context, clips, sample_times, num_instances, device_params, poses, pose_stride_bytes, hip_stream);
	ACLHIP_CHECK_HIP(context, hipEventRecord(stop, hip_stream));
	ACLHIP_CHECK_HIP(context, hipEventSynchronize(stop));
	float elapsed_ms = 0.0f;
	ACLHIP_CHECK_HIP(context, hipEventElapsedTime(&elapsed_ms, start, stop));
	(void)hipEventDestroy(start);
	(void)hipEventDestroy(stop);
	*out_ms_per_launch = elapsed_ms / float(repeats);
	return status;
}

extern "C" aclhip_status aclhip_time_decompress_poses_batch(aclhip_context* context, const aclhip_clip* clips, const float* sample_times, uint32_t num_instances,
	const aclhip_decompress_params* params, const aclhip_pose_consumers* consumers, void* poses, uint64_t pose_stride_bytes, void* stream, uint32_t repeats, float* out_ms_per_launch)
{
	if (out_ms_per_launch == nullptr || repeats == 0 || consumers == nullptr)
		return ACLHIP_ERROR_INVALID_ARGUMENT;
	aclhip_status status = check_batch_arguments(context, clips, sample_times, num_instances, poses, pose_stride_bytes);
	if (status != ACLHIP_OK)
		return status;

	decode_params device_params;
	status = resolve_params(context, params, device_params);
	if (status != ACLHIP_OK)
		return status;

	device_guard guard(context->device);
	hipStream_t hip_stream = static_cast<hipStream_t>(stream);
	hipEvent_t start, stop;
	ACLHIP_CHECK_HIP(context, hipEventCreate(&start));
	ACLHIP_CHECK_HIP(context, hipEventCreate(&stop));
	ACLHIP_CHECK_HIP(context, hipEventRecord(start, hip_stream));
	for (uint32_t i = 0; i < repeats && status == ACLHIP_OK; ++i)
		status = launch_consumers(context, clips, sample_times, num_instances, device_params, *consumers, poses, pose_stride_bytes, hip_stream);
	ACLHIP_CHECK_HIP(context, hipEventRecord(stop, hip_stream));
	ACLHIP_CHECK_HIP(context, hipEventSynchronize(stop));
	float elapsed_ms = 0.0f;
	ACLHIP_CHECK_HIP(context, hipEventElapsedTime(&elapsed_ms, start, stop));
	(void)hipEventDestroy(start);
	(void)hipEventDestroy(stop);
	*out_ms_per_launch = elapsed_ms / float(repeats);
	return status;
}

extern "C" aclhip_status aclhip_describe_tracks_launch(aclhip_context* context, const aclhip_decompress_params* params, const aclhip_output_desc* output, uint64_t pose_stride_bytes,
	char* out_name, uint32_t capacity, uint32_t* out_windows_per_instance)
{
	if (context == nullptr || out_name == nullptr || capacity == 0)
		return ACLHIP_ERROR_INVALID_ARGUMENT;
	decode_params device_params;
	aclhip_status status = resolve_params(context, params, device_params);
	if (status == ACLHIP_OK)
		status = apply_output_desc(context, output, device_params);
	if (status != ACLHIP_OK)
		return status;
	// the very functions launch_tracks asks (shape from the batch's stride, kernel from shape and settings: ACLHIP_WIDE_KEY_LOADS and
	// ACLHIP_FORCE_GENERIC_KERNEL included)
	std::lock_guard<std::shared_mutex> lock(context->mutex);
	const pose_launch_shape shape = pose_launch_shape_of(context, device_params.layout, pose_stride_bytes);
	const char* name = "";
	(void)pose_kernel_of(context, device_params, shape, &name);
	std::snprintf(out_name, capacity, "%s", name);
	if (out_windows_per_instance != nullptr)
		*out_windows_per_instance = shape.windows_per_instance;
	return ACLHIP_OK;
}

extern "C" aclhip_status aclhip_describe_tracks_kernel(aclhip_context* context, const aclhip_decompress_params* params, char* out_name, uint32_t capacity)
{
	// rows as wide as the largest registered clip: what a caller that sizes its pose buffer from the registry launches
	return aclhip_describe_tracks_launch(context, params, nullptr, ~uint64_t(0), out_name, capacity, nullptr);
}

extern "C" aclhip_status aclhip_measure_pose_store_bandwidth(aclhip_context* context, void* poses, uint64_t pose_stride_bytes, uint32_t num_instances, uint32_t num_tracks,
	uint32_t repeats, void* stream, float* out_gb_per_second, uint32_t* out_waves_per_cu)
{
	if (context == nullptr || poses == nullptr || out_gb_per_second == nullptr || repeats == 0 || num_instances == 0 || num_tracks == 0
		|| (reinterpret_cast<uintptr_t>(poses) & 15u) != 0 || (pose_stride_bytes & 15u) != 0 || pose_stride_bytes < uint64_t(num_tracks) * 48)
		return ACLHIP_ERROR_INVALID_ARGUMENT;

	device_guard guard(context->device);
	hipStream_t hip_stream = static_cast<hipStream_t>(stream);
	const uint32_t pose_quads = num_tracks * 3;
	const uint32_t windows = (pose_quads + k_image_chunk_quads - 1) / k_image_chunk_quads;
	const uint64_t num_waves = uint64_t(num_instances) * windows;
	if (num_waves > 0xFFFFFFF0ull)
		return ACLHIP_ERROR_INVALID_ARGUMENT;
	const uint32_t num_blocks = uint32_t((num_waves + k_waves_per_block - 1) / k_waves_per_block);
	ACLHIP_CHECK_HIP(context, hipFuncSetAttribute(reinterpret_cast<const void*>(pose_store_stream_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256));
	// (released on every way out)
	struct probe_resources
	{
		hipEvent_t start = nullptr, stop = nullptr;
		uint32_t* chain = nullptr;
		~probe_resources()
		{
			if (start != nullptr) (void)hipEventDestroy(start);
			if (stop != nullptr) (void)hipEventDestroy(stop);
			if (chain != nullptr) (void)hipFree(chain);
		}
	} resources;
	ACLHIP_CHECK_HIP(context, hipEventCreate(&resources.start));
	ACLHIP_CHECK_HIP(context, hipEventCreate(&resources.stop));
	const hipEvent_t start = resources.start, stop = resources.stop;
	{
		// the probe launches on the caller's stream like any decode: the context has to know the stream (retired clips wait for it)
		std::lock_guard<std::shared_mutex> lock(context->mutex);
		note_launch_stream(context, hip_stream);
	}
	// workgroups of 4 waves per CU by the LDS a workgroup asks for: 20 KB -> 8 (32 waves), 40 KB -> 4, 52 KB -> 3, 80 KB -> 2 (8 waves);
	// 0 / 3 / 6 dependent scalar loads in front of the stores (a 16 KB table of indices, resident in the L2s)
	{
		std::vector<uint32_t> host_chain(4096);
		for (uint32_t i = 0; i < 4096; ++i)
			host_chain[i] = (i * 1237u + 511u) & 4095u;
		ACLHIP_CHECK_HIP(context, hipMalloc(reinterpret_cast<void**>(&resources.chain), host_chain.size() * sizeof(uint32_t)));
		if (hipMemcpy(resources.chain, host_chain.data(), host_chain.size() * sizeof(uint32_t), hipMemcpyHostToDevice) != hipSuccess)
			return fail(context, ACLHIP_ERROR_DEVICE, "uploading the probe's table failed");
	}
	uint32_t* const chain = resources.chain;
	const uint32_t lds_bytes[4] = { 20 * 1024, 40 * 1024, 52 * 1024, 80 * 1024 };
	const uint32_t waves_per_cu[4] = { 32, 16, 12, 8 };
	float best = 0.0f;
	for (uint32_t shape = 0; shape < 4; ++shape)
		for (uint32_t hops = 0; hops <= 6; hops += 3)
		{
			hipLaunchKernelGGL(pose_store_stream_kernel, dim3(num_blocks), dim3(k_block_size), lds_bytes[shape], hip_stream, static_cast<uint8_t*>(poses), pose_stride_bytes, num_instances, pose_quads, windows, chain, hops, 0.0f);
			(void)hipEventRecord(start, hip_stream);
			for (uint32_t i = 0; i < repeats; ++i)
				hipLaunchKernelGGL(pose_store_stream_kernel, dim3(num_blocks), dim3(k_block_size), lds_bytes[shape], hip_stream, static_cast<uint8_t*>(poses), pose_stride_bytes, num_instances, pose_quads, windows, chain, hops, float(i));
			(void)hipEventRecord(stop, hip_stream);
			float elapsed_ms = 0.0f;
			if (hipEventSynchronize(stop) != hipSuccess || hipEventElapsedTime(&elapsed_ms, start, stop) != hipSuccess)
				return fail(context, ACLHIP_ERROR_DEVICE, "the store stream probe failed");
			const float rate = float(double(num_instances) * pose_quads * 16.0 * repeats / (double(elapsed_ms) * 1.0e-3) / 1.0e9);
			if (rate > best)
			{
				best = rate;
				if (out_waves_per_cu != nullptr)
					*out_waves_per_cu = waves_per_cu[shape];
			}
		}
	// ... and the runtime's own fill of the same bytes (hipMemsetAsync: few waves, each sweeping a large contiguous range): the decode
	// of one-window poses, paced by its seek, comes out ahead of every pose shaped store-only launch above, not of this one
	{
		const size_t fill_bytes = size_t(num_instances - 1) * pose_stride_bytes + size_t(pose_quads) * 16;
		(void)hipMemsetAsync(poses, 0, fill_bytes, hip_stream);
		(void)hipEventRecord(start, hip_stream);
		for (uint32_t i = 0; i < repeats; ++i)
			(void)hipMemsetAsync(poses, int(i & 1), fill_bytes, hip_stream);
		(void)hipEventRecord(stop, hip_stream);
		float elapsed_ms = 0.0f;
		if (hipEventSynchronize(stop) == hipSuccess && hipEventElapsedTime(&elapsed_ms, start, stop) == hipSuccess && elapsed_ms > 0.0f)
		{
			const float rate = float(double(fill_bytes) * repeats / (double(elapsed_ms) * 1.0e-3) / 1.0e9);
			if (rate > best)
			{
				best = rate;
				if (out_waves_per_cu != nullptr)
					*out_waves_per_cu = 0;		// the runtime's fill kernel
			}
		}
	}
	*out_gb_per_second = best;
	return ACLHIP_OK;
}

extern "C" aclhip_status aclhip_measure_write_bandwidth(aclhip_context* context, void* buffer, uint64_t size_bytes, uint32_t repeats, void* stream, float* out_gb_per_second)
{
	if (context == nullptr || buffer == nullptr || out_gb_per_second == nullptr || repeats == 0 || size_bytes < 16 || (reinterpret_cast<uintptr_t>(buffer) & 15u) != 0)
		return ACLHIP_ERROR_INVALID_ARGUMENT;

	device_guard guard(context->device);
	hipStream_t hip_stream = static_cast<hipStream_t>(stream);
	const uint64_t num_quads = size_bytes / 16;
	const uint32_t num_blocks = uint32_t(std::min<uint64_t>((num_quads + k_block_size - 1) / k_block_size, 256ull * 32ull));
	hipEvent_t start, stop;
	ACLHIP_CHECK_HIP(context, hipEventCreate(&start));
	ACLHIP_CHECK_HIP(context, hipEventCreate(&stop));
	hipLaunchKernelGGL(stream_write_kernel, dim3(num_blocks), dim3(k_block_size), 0, hip_stream, static_cast<float4*>(buffer), num_quads, 0.0f);
	ACLHIP_CHECK_HIP(context, hipEventRecord(start, hip_stream));
	for (uint32_t i = 0; i < repeats; ++i)
		hipLaunchKernelGGL(stream_write_kernel, dim3(num_blocks), dim3(k_block_size), 0, hip_stream, static_cast<float4*>(buffer), num_quads, float(i));
	ACLHIP_CHECK_HIP(context, hipEventRecord(stop, hip_stream));
	ACLHIP_CHECK_HIP(context, hipEventSynchronize(stop));
	float elapsed_ms = 0.0f;
	ACLHIP_CHECK_HIP(context, hipEventElapsedTime(&elapsed_ms, start, stop));
	(void)hipEventDestroy(start);
	(void)hipEventDestroy(stop);
	*out_gb_per_second = float(double(num_quads) * 16.0 * repeats / (double(elapsed_ms) * 1.0e-3) / 1.0e9);
	return ACLHIP_OK;
}

extern "C" aclhip_status aclhip_batch_algorithmic_bytes(const aclhip_context* context, const aclhip_clip* clips, uint32_t num_instances,
	uint64_t* out_bytes_written, uint64_t* out_distinct_clip_bytes)
{
	if (context == nullptr || (num_instances != 0 && clips == nullptr) || out_bytes_written == nullptr || out_distinct_clip_bytes == nullptr)
		return ACLHIP_ERROR_INVALID_ARGUMENT;

	std::lock_guard<std::shared_mutex> lock(const_cast<aclhip_context*>(context)->mutex);
	std::unordered_set<uint32_t> distinct;
	uint64_t written = 0, read = 0;
	for (uint32_t i = 0; i < num_instances; ++i)
	{
		const uint32_t clip = clips[i];
		if (clip >= context->clips.size() || !context->clips[clip].in_use)
			continue;
		written += uint64_t(context->clips[clip].info.num_tracks) * context->clips[clip].info.num_components * 4;		// 48 bytes per transform track
		if (distinct.insert(clip).second)
			read += context->clips[clip].touched_bytes;
	}
	*out_bytes_written = written;
	*out_distinct_clip_bytes = read;
	return ACLHIP_OK;
}
