// host_lists.inl -- part of aclhip.hip (one translation unit; included there, in this order, not compiled on its own).
// Host side: persistent instance lists (include/aclhip.h: aclhip_instance_list_*). The library owns the decode order of a list whose
// clip assignment outlives a frame: ordered once (order_instances_on_device), patched in place when a few instances change clip
// (update_instance_list_kernel), re-ordered before a decode once an eighth of it has changed.

struct aclhip_context::instance_list
{
	bool in_use = false;
	uint32_t num_instances = 0;
	uint32_t* d_memory = nullptr;			// clips (instance order) | order (slot -> instance) | positions (instance -> slot) | ordered clips (slot order)
	uint32_t changed_since_ordered = 0;		// instances whose clip changed since the list was last ordered
	bool ordered = false;
	uint32_t ordered_for_windows = 0;		// waves per pose of the launch shape the order was made for (the slot -> XCD map depends on it); 0: the registry's at that time
	uint64_t num_orderings = 0;
	const uint32_t* attached_clips = nullptr;	// aclhip_instance_list_attach: the caller's clip array (instance order) the list decodes, or null (the list's own copy)

	uint32_t* clips() const { return d_memory; }
	uint32_t* order() const { return d_memory + num_instances; }
	uint32_t* positions() const { return d_memory + size_t(num_instances) * 2; }
	uint32_t* ordered_clips() const { return d_memory + size_t(num_instances) * 3; }
};

void free_instance_lists(aclhip_context* context)
{
	for (aclhip_context::instance_list& list : context->instance_lists)
		if (list.in_use && list.d_memory != nullptr)
			(void)hipFree(list.d_memory);
	context->instance_lists.clear();
}

namespace
{
	aclhip_context::instance_list* find_list(aclhip_context* context, aclhip_instance_list list)
	{
		return list < context->instance_lists.size() && context->instance_lists[list].in_use ? &context->instance_lists[list] : nullptr;
	}

	// An ordering kernel of the one launch form gave up on this stream and the host has not reported it yet (order_grid_barrier,
	// kernels_misc.inl: every ordering queued on the stream since then wrote the identity order -- valid and consistent, without
	// locality). A peek at pinned host memory, no synchronization. (the caller holds the registry lock)
	bool ordering_gave_up_on(const aclhip_context* context, hipStream_t stream)
	{
		for (const aclhip_context::order_scratch& scratch : context->order_scratches)
			if (scratch.stream == stream && scratch.host_failed != nullptr && __atomic_load_n(scratch.host_failed, __ATOMIC_RELAXED) != 0)
				return true;
		return false;
	}
}

extern "C" aclhip_status aclhip_instance_list_create(aclhip_context* context, uint32_t num_instances, aclhip_instance_list* out_list)
{
	if (context == nullptr || out_list == nullptr)
		return ACLHIP_ERROR_INVALID_ARGUMENT;
	*out_list = ACLHIP_INVALID_HANDLE;
	if (num_instances == 0)
		return fail(context, ACLHIP_ERROR_INVALID_ARGUMENT, "an instance list holds at least one instance");
	device_guard guard(context->device);
	uint32_t* memory = nullptr;
	ACLHIP_CHECK_HIP(context, hipMalloc(reinterpret_cast<void**>(&memory), size_t(num_instances) * 4 * sizeof(uint32_t)));

	std::lock_guard<std::shared_mutex> lock(context->mutex);
	uint32_t slot = 0;
	while (slot < context->instance_lists.size() && context->instance_lists[slot].in_use)
		slot++;
	if (slot == context->instance_lists.size())
		context->instance_lists.emplace_back();
	aclhip_context::instance_list& list = context->instance_lists[slot];
	list = aclhip_context::instance_list();
	list.in_use = true;
	list.num_instances = num_instances;
	list.d_memory = memory;
	*out_list = slot;
	return ACLHIP_OK;
}

extern "C" aclhip_status aclhip_instance_list_destroy(aclhip_context* context, aclhip_instance_list handle)
{
	if (context == nullptr)
		return ACLHIP_ERROR_INVALID_ARGUMENT;
	std::lock_guard<std::shared_mutex> lock(context->mutex);
	aclhip_context::instance_list* list = find_list(context, handle);
	if (list == nullptr)
		return fail(context, ACLHIP_ERROR_INVALID_ARGUMENT, "unknown instance list %u", handle);
	device_guard guard(context->device);
	// decodes of the list may still be in flight: its memory is retired behind them
	aclhip_context::retired_item item;
	item.device_memory = list->d_memory;
	retire(context, std::move(item));
	*list = aclhip_context::instance_list();
	return ACLHIP_OK;
}

extern "C" aclhip_status aclhip_instance_list_set_clips(aclhip_context* context, aclhip_instance_list handle, const aclhip_clip* clips, void* stream)
{
	if (context == nullptr)
		return ACLHIP_ERROR_INVALID_ARGUMENT;
	if (clips == nullptr)
		return fail(context, ACLHIP_ERROR_INVALID_ARGUMENT, "null clip list");
	aclhip_context::instance_list snapshot;
	{
		std::lock_guard<std::shared_mutex> lock(context->mutex);
		aclhip_context::instance_list* list = find_list(context, handle);
		if (list == nullptr)
			return fail(context, ACLHIP_ERROR_INVALID_ARGUMENT, "unknown instance list %u", handle);
		// the clips are about to be overwritten: whatever order the list had describes them no longer. It counts as ordered again
		// only once the new order has been enqueued (a failure below leaves a list that refuses decodes, not one that decodes a stale assignment)
		list->ordered = false;
		list->attached_clips = nullptr;
		snapshot = *list;
	}
	device_guard guard(context->device);
	ACLHIP_CHECK_HIP(context, hipMemcpyAsync(snapshot.clips(), clips, size_t(snapshot.num_instances) * sizeof(uint32_t), hipMemcpyDeviceToDevice, static_cast<hipStream_t>(stream)));
	// (the shape of the decodes to come is not known yet: ordered for the largest registered clip, again by the first decode whose pose
	// stride says otherwise)
	// (what "the largest registered clip" means is settled HERE: a clip registered or unregistered before the first decode must not
	// make that decode mistake the order for one of its own shape)
	const uint32_t registry_windows = windows_per_instance_of(context);		// (takes the context's lock itself)
	const aclhip_status status = order_instances_on_device(context, registry_windows, snapshot.clips(), nullptr, snapshot.num_instances, snapshot.order(), snapshot.ordered_clips(), nullptr, snapshot.positions(), stream);		// (takes the context's lock itself)
	if (status != ACLHIP_OK)
		return status;
	std::lock_guard<std::shared_mutex> lock(context->mutex);
	aclhip_context::instance_list* list = find_list(context, handle);
	if (list != nullptr && list->d_memory == snapshot.d_memory)
	{
		list->ordered = true;
		list->ordered_for_windows = registry_windows;
		list->changed_since_ordered = 0;
		list->num_orderings++;
	}
	return ACLHIP_OK;
}

extern "C" aclhip_status aclhip_instance_list_attach(aclhip_context* context, aclhip_instance_list handle, const aclhip_clip* caller_clips, void* stream)
{
	if (context == nullptr)
		return ACLHIP_ERROR_INVALID_ARGUMENT;
	if (caller_clips == nullptr)
		return fail(context, ACLHIP_ERROR_INVALID_ARGUMENT, "null clip list");
	aclhip_context::instance_list snapshot;
	{
		std::lock_guard<std::shared_mutex> lock(context->mutex);
		aclhip_context::instance_list* list = find_list(context, handle);
		if (list == nullptr)
			return fail(context, ACLHIP_ERROR_INVALID_ARGUMENT, "unknown instance list %u", handle);
		list->ordered = false;		// (see aclhip_instance_list_set_clips)
		list->attached_clips = nullptr;
		snapshot = *list;
	}
	device_guard guard(context->device);
	const uint32_t registry_windows = windows_per_instance_of(context);		// (takes the context's lock itself)
	// the order of the caller's array as it is now; no copy of the clips, no slot-order copy, no positions: the decode reads the caller's array through the order
	const aclhip_status status = order_instances_on_device(context, registry_windows, caller_clips, nullptr, snapshot.num_instances, snapshot.order(), nullptr, nullptr, nullptr, stream);
	if (status != ACLHIP_OK)
		return status;
	std::lock_guard<std::shared_mutex> lock(context->mutex);
	aclhip_context::instance_list* list = find_list(context, handle);
	if (list != nullptr && list->d_memory == snapshot.d_memory)
	{
		list->ordered = true;
		list->attached_clips = caller_clips;
		list->ordered_for_windows = registry_windows;
		list->changed_since_ordered = 0;
		list->num_orderings++;
	}
	return ACLHIP_OK;
}

extern "C" aclhip_status aclhip_instance_list_note_changes(aclhip_context* context, aclhip_instance_list handle, uint32_t count)
{
	if (context == nullptr)
		return ACLHIP_ERROR_INVALID_ARGUMENT;
	std::lock_guard<std::shared_mutex> lock(context->mutex);
	aclhip_context::instance_list* list = find_list(context, handle);
	if (list == nullptr)
		return fail(context, ACLHIP_ERROR_INVALID_ARGUMENT, "unknown instance list %u", handle);
	if (!list->ordered || list->attached_clips == nullptr)
		return fail(context, ACLHIP_ERROR_INVALID_ARGUMENT, "aclhip_instance_list_attach comes first");
	list->changed_since_ordered = uint32_t(std::min<uint64_t>(uint64_t(list->changed_since_ordered) + count, 0xFFFFFFFFull));
	return ACLHIP_OK;
}

extern "C" aclhip_status aclhip_instance_list_update(aclhip_context* context, aclhip_instance_list handle, const uint32_t* instances, const aclhip_clip* clips, uint32_t count, void* stream)
{
	if (context == nullptr)
		return ACLHIP_ERROR_INVALID_ARGUMENT;
	if (count == 0)
		return ACLHIP_OK;
	if (instances == nullptr || clips == nullptr)
		return fail(context, ACLHIP_ERROR_INVALID_ARGUMENT, "null update lists");
	std::lock_guard<std::shared_mutex> lock(context->mutex);
	aclhip_context::instance_list* list = find_list(context, handle);
	if (list == nullptr)
		return fail(context, ACLHIP_ERROR_INVALID_ARGUMENT, "unknown instance list %u", handle);
	if (!list->ordered)
		return fail(context, ACLHIP_ERROR_INVALID_ARGUMENT, "aclhip_instance_list_set_clips comes first");
	if (list->attached_clips != nullptr)
		return fail(context, ACLHIP_ERROR_INVALID_ARGUMENT, "an attached list decodes the caller's own clip array: write the changes there and call aclhip_instance_list_note_changes");
	device_guard guard(context->device);
	note_launch_stream(context, static_cast<hipStream_t>(stream));
	hipLaunchKernelGGL(update_instance_list_kernel, dim3((count + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream),
		instances, clips, count, list->num_instances, list->clips(), list->positions(), list->ordered_clips());
	ACLHIP_CHECK_HIP(context, hipGetLastError());
	list->changed_since_ordered = uint32_t(std::min<uint64_t>(uint64_t(list->changed_since_ordered) + count, 0xFFFFFFFFull));
	return ACLHIP_OK;
}

extern "C" aclhip_status aclhip_decompress_tracks_list(aclhip_context* context, aclhip_instance_list handle, const float* sample_times, const aclhip_decompress_params* params,
	const aclhip_output_desc* output, int poses_in_instance_order, void* poses, uint64_t pose_stride_bytes, void* stream)
{
	if (context == nullptr)
		return ACLHIP_ERROR_INVALID_ARGUMENT;
	// everything that can refuse the call comes first: the list's bookkeeping changes only when its (re-)ordering has been enqueued
	if (output != nullptr && output->rows != nullptr)
		return fail(context, ACLHIP_ERROR_INVALID_ARGUMENT, "an instance list decides the rows itself (poses_in_instance_order)");
	decode_params device_params;
	aclhip_status status = resolve_params(context, params, device_params);
	if (status == ACLHIP_OK)
		status = apply_output_desc(context, output, device_params);
	if (status != ACLHIP_OK)
		return status;

	aclhip_context::instance_list snapshot;
	bool reorder = false;
	uint32_t windows_per_instance = 1;
	bool ordering_gave_up = false;
	{
		std::lock_guard<std::shared_mutex> lock(context->mutex);
		aclhip_context::instance_list* list = find_list(context, handle);
		if (list == nullptr)
			return fail(context, ACLHIP_ERROR_INVALID_ARGUMENT, "unknown instance list %u", handle);
		if (!list->ordered)
			return fail(context, ACLHIP_ERROR_INVALID_ARGUMENT, "aclhip_instance_list_set_clips comes first");
		snapshot = *list;
		windows_per_instance = pose_launch_shape_of(context, device_params.layout, pose_stride_bytes).windows_per_instance;
		ordering_gave_up = ordering_gave_up_on(context, static_cast<hipStream_t>(stream));
		// an ordering on this stream gave up: this list's order may be the identity. The ordering below reports it (once, and this decode
		// is refused with it); the decode after that orders the list again, with the form that cannot give up
		if (ordering_gave_up)
			list->changed_since_ordered = list->num_instances;
	}
	const bool attached = snapshot.attached_clips != nullptr;
	status = check_batch_arguments(context, snapshot.clips(), sample_times, snapshot.num_instances, poses, pose_stride_bytes);
	if (status != ACLHIP_OK)
		return status;

	// enough of the list plays other clips than when it was ordered, or it was ordered for launches of another shape (which slot of a
	// launch runs on which XCD follows from the waves a pose takes): order it again, in front of this decode
	static const uint32_t reorder_divisor = []() { const char* value = lab_knob("ACLHIP_LIST_REORDER_DIVISOR"); return value != nullptr ? uint32_t(std::max(1L, std::atol(value))) : 8u; }();		// (measurement knob)
	reorder = ordering_gave_up || uint64_t(snapshot.changed_since_ordered) * reorder_divisor >= snapshot.num_instances || snapshot.ordered_for_windows != windows_per_instance;

	device_params.time_indices = snapshot.order();
	device_params.instance_rows = poses_in_instance_order != 0 ? snapshot.order() : nullptr;
	device_params.clips_by_caller_instance = attached ? 1 : 0;

	device_guard guard(context->device);
	if (reorder)
	{
		if (attached)
			status = order_instances_on_device(context, windows_per_instance, snapshot.attached_clips, nullptr, snapshot.num_instances, snapshot.order(), nullptr, nullptr, nullptr, stream);
		else
			status = order_instances_on_device(context, windows_per_instance, snapshot.clips(), nullptr, snapshot.num_instances, snapshot.order(), snapshot.ordered_clips(), nullptr, snapshot.positions(), stream);
		if (status != ACLHIP_OK)
			return status;
		std::lock_guard<std::shared_mutex> lock(context->mutex);
		aclhip_context::instance_list* list = find_list(context, handle);
		if (list != nullptr && list->d_memory == snapshot.d_memory)
		{
			// (updates that arrived between the snapshot and here are part of what was just ordered or will count towards the next time)
			list->changed_since_ordered -= std::min(list->changed_since_ordered, snapshot.changed_since_ordered);
			list->ordered_for_windows = windows_per_instance;
			list->num_orderings++;
		}
	}
	return launch_tracks(context, attached ? snapshot.attached_clips : snapshot.ordered_clips(), sample_times, snapshot.num_instances, device_params, poses, pose_stride_bytes, static_cast<hipStream_t>(stream));
}

extern "C" aclhip_status aclhip_instance_list_get_order(aclhip_context* context, aclhip_instance_list handle, const uint32_t** out_order, uint64_t* out_num_orderings)
{
	if (context == nullptr)
		return ACLHIP_ERROR_INVALID_ARGUMENT;
	std::lock_guard<std::shared_mutex> lock(context->mutex);
	aclhip_context::instance_list* list = find_list(context, handle);
	if (list == nullptr)
		return fail(context, ACLHIP_ERROR_INVALID_ARGUMENT, "unknown instance list %u", handle);
	if (out_order != nullptr)
		*out_order = list->order();
	if (out_num_orderings != nullptr)
		*out_num_orderings = list->num_orderings;
	return ACLHIP_OK;
}
