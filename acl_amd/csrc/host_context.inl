// host_context.inl -- part of aclhip.hip (one translation unit; included there, in this order, not compiled on its own).
// Host side: registries, the context, clip memory slabs, errors, blob validation, parameters, create / destroy.


namespace
{
	struct host_clip
	{
		bool in_use = false;
		uint32_t database = ACLHIP_INVALID_HANDLE;
		void* device_memory = nullptr;		// one piece of a slab: blob | base pose | quad map | animated tracks
		uint32_t* d_hierarchy = nullptr;	// aclhip_set_clip_hierarchy
		aclhip_clip_info info = {};
		uint64_t touched_bytes = 0;			// bytes of the blob + tables a decode may read
		// what this clip asks of a launch (the context keeps the maxima over its live clips)
		uint32_t pose_quads = 0, hierarchy_words = 0, scalar_tracks = 0, scalar_frame_bytes = 0;
		uint32_t window_animated = 0, window_key_bytes = 0;		// staged kernel: animated sub-tracks of the fullest pose window, LDS bytes to stage one keyframe's runs
		bool scaled = false;				// qvvf clip with scale sub-tracks, or whose default scale is not 1
		bool negative_scale = false;		// some scale sub-track may decode a negative component (mirrored rigs): rtm::qvv_mul then goes through matrices
		bool wide_scalar = false;			// scalar track list of more than one float per track
		uint32_t db_first_segment_header = 0, db_num_segments = 0;	// bound to a streamed database: its runtime segment headers (host_database::segment_pose_bits)
		// the blob's optional metadata, read at registration (parse_clip_metadata, host_clips.inl)
		aclhip_clip_metadata_info metadata = {};
		std::vector<uint32_t> metadata_parents;			// [num_tracks] when has_parent_track_indices
		std::vector<float> metadata_descriptions;		// [num_tracks][14]: default_value as 12 floats, precision, shell_distance, when has_track_descriptions
	};
}

namespace
{
	struct host_database
	{
		bool in_use = false;
		aclhip_database_info info = {};
		uint32_t hash = 0;
		uint32_t num_bound_clips = 0;
		uint64_t runtime_headers_size = 0;
		uint8_t* d_runtime_headers = nullptr;			// [clip header, segment headers...] per clip, zeroed = nothing streamed in
		uint8_t* d_bulk_data[2] = { nullptr, nullptr };	// HBM residence of each tier
		uint8_t* pinned_bulk_data[2] = { nullptr, nullptr };	// the streamer's backing store (hipHostMalloc)
		tier_patch* d_patches[2] = { nullptr, nullptr };
		std::vector<database_chunk_description> chunks[2];
		std::vector<uint32_t> chunk_first_patch[2];		// num_chunks + 1 entries
		std::vector<uint32_t> loaded_chunks[2];			// bitset, first chunk in the MSB like core/bitset.h
		std::vector<database_clip_metadata> clip_metadata;
		std::vector<tier_patch> patches_by_header[2];	// every (known) chunk segment of the tier, sorted by segment_header_offset (bounds checks when a clip is bound)
		// databases whose bulk data arrives through the caller's own streamers (aclhip_register_database_streamed): chunks are parsed
		// and checked when they first arrive, their patches appended in chunk order
		bool streamed = false;
		uint32_t num_parsed_chunks[2] = { 0, 0 };
		uint32_t num_uploaded_patches[2] = { 0, 0 };	// patches of the parsed chunks whose copy to d_patches has been enqueued
		tier_patch* pinned_patches[2] = { nullptr, nullptr };	// host mirror of d_patches the uploads are made from
		uint32_t patch_capacity[2] = { 0, 0 };
		std::vector<std::pair<uint32_t, uint32_t>> segment_pose_bits;	// (runtime segment header offset, bits per keyframe) of the bound clips
		// the clips bound to the database, for refresh_database_sample_tiers_kernel: host list and its device copy
		std::vector<uint32_t> bound_clips;
		uint32_t* d_bound_clips = nullptr;
		uint32_t bound_clips_capacity = 0;
	};
}

struct aclhip_context
{
	int device = 0;
	// The registry lock. Everything that registers, retires, orders or changes a list holds it exclusively; the LAUNCHES (pose, track,
	// scalar, consumer batches) hold it SHARED across their enqueue: decoding threads of one context no longer serialize on it (until
	// round 5 they did), while an unregistration still cannot record its "everything enqueued so far" events in the middle of a launch
	// that has noted its stream but not yet enqueued its kernel.
	std::shared_mutex mutex;
	std::mutex streams_mutex;				// launch_streams, among launching threads (they only hold `mutex` shared)
	std::vector<host_clip> clips;
	std::vector<uint32_t> free_slots;
	std::vector<host_database> databases;
	device_clip* d_clips = nullptr;
	uint32_t d_clips_capacity = 0;
	unsigned long long* d_rejected = nullptr;	// [0] instances the kernels refused, [1] transforms of the pose consumers that met a negative scale
	uint32_t max_pose_quads = 0;			// largest pose (3 * num_tracks) among registered clips
	uint32_t max_window_animated = 0;		// most animated sub-tracks in one pose window among registered clips
	uint32_t max_window_key_bytes = 0;		// most LDS bytes staging one keyframe's runs of a window takes, among registered clips
	uint32_t num_wide_scalar_clips = 0;		// live scalar track lists of 2 - 4 floats per track (while 0 the grouped scalar kernel is compiled for float1f only)
	uint32_t num_scaled_clips = 0;			// live clips whose scale is not 1 everywhere (the pose consumers keep no scale in LDS while this is 0)
	uint32_t num_negative_scale_clips = 0;	// live clips that may decode a negative scale (while 0 the pose consumers are compiled without rtm::qvv_mul's matrix route)
	uint32_t max_hierarchy_words = 0;		// largest walk schedule (aclhip_set_clip_hierarchy) among registered clips
	uint32_t max_scalar_tracks = 0;			// largest scalar track list among registered clips
	uint32_t max_scalar_frame_bytes = 0;	// largest frame (one sample of every track) among registered scalar clips
	uint32_t num_compute_units = 256;		// of the device (MI355X: 256): the persistent kernels fill it once
	bool force_generic_kernel = false;		// testing aid (ACLHIP_FORCE_GENERIC_KERNEL=1): always launch the any-settings kernel

	// Clips live in a few large HBM slabs instead of one hipMalloc each: a batch that draws on hundreds of clips then touches a
	// handful of large, contiguously mapped regions (fewer address translations to miss) and registration stops paying for an
	// allocation per clip. Bump allocation inside a slab; freeing rolls the bump pointer back over every freed piece at the top,
	// and a slab is recycled when its last clip is unregistered.
	// Walk schedules (aclhip_set_clip_hierarchy), one image per distinct hierarchy: clips of one skeleton share it, which is also
	// what lets a workgroup whose instances share a skeleton keep a single copy in LDS
	struct hierarchy_image { std::vector<uint32_t> parents; uint32_t* d_image = nullptr; uint32_t num_users = 0; };
	std::vector<hierarchy_image> hierarchies;

	// ---- stream ordered clip lifetime --------------------------------------------------------------------------------------------
	// Registering, replacing and unregistering clips never stalls the device: uploads go through a pinned staging buffer on the
	// context's own NON-BLOCKING copy stream (only the calling thread waits for its copy), and what an unregistration retires is freed
	// once an event recorded on every stream this context has launched on has completed -- polled by later calls, never waited for
	// (except in aclhip_destroy). The clip table never moves: its address range is reserved up front and backed page by page.
	hipStream_t copy_stream = nullptr;
	hipStream_t retire_stream = nullptr;		// clears the table records of unregistered clips BEHIND the work in flight (retire)
	uint8_t* pinned_staging = nullptr;
	size_t pinned_staging_bytes = 0;
	std::vector<hipStream_t> launch_streams;		// every stream work was enqueued on (nullptr = the default stream)
	std::vector<hipEvent_t> event_pool;
	struct retired_item
	{
		std::vector<hipEvent_t> events;				// one per launch stream, recorded when the item was retired
		void* clip_memory = nullptr;				// a piece of a slab
		uint32_t* hierarchy = nullptr;				// a walk schedule image (shared images are reference counted)
		uint32_t slot = ACLHIP_INVALID_HANDLE;		// clip handle that becomes reusable
		uint8_t* database_memory[6] = { nullptr, nullptr, nullptr, nullptr, nullptr, nullptr };		// hipMalloc'ed pieces of a database
		uint8_t* database_pinned[4] = { nullptr, nullptr, nullptr, nullptr };		// bulk data x 2, patch mirrors x 2
		void* device_memory = nullptr;				// any other hipMalloc'ed piece
	};
	std::vector<retired_item> retired;
	// aclhip_order_instances_device: per stream, the per clip counters (zero between calls) followed by the per clip cursors
	struct order_scratch
	{
		hipStream_t stream = nullptr;
		uint32_t* bins = nullptr;
		size_t capacity = 0;						// words allocated
		size_t zeroed_bins = 0;						// the counters | cursors layout (padded bins per half) the words were zeroed for; 0: not zeroed
		uint32_t* barrier = nullptr;				// the one launch form's barrier words (order_control), zeroed when allocated
		uint32_t* host_failed = nullptr;			// pinned, device visible: a kernel of the one launch form gave up at a barrier (order_grid_barrier)
		bool one_launch_form_disabled = false;		// ... after which this stream orders with the three launch form
	};
	std::vector<order_scratch> order_scratches;
	// aclhip_instance_list_*: instance lists kept in decode order (host_lists.inl)
	struct instance_list;
	std::vector<instance_list> instance_lists;
	uint64_t clips_registered = 0;					// statistics (aclhip_get_lifetime_stats)
	uint64_t clips_unregistered = 0;
	uint64_t deferred_frees_completed = 0;

	// the clip table: reserved address range, mapped in `table_granularity` steps (virtual memory management API); when the device does
	// not offer that, one fixed allocation of k_fixed_table_entries records
	bool table_is_virtual = false;
	size_t table_reserved_bytes = 0;
	size_t table_mapped_bytes = 0;
	size_t table_granularity = 0;
	std::vector<hipMemGenericAllocationHandle_t> table_handles;

	// buffers of other processes mapped by aclhip_peer_open_buffer: the pointer handed out and the allocation it lies in
	struct peer_mapping { void* buffer; void* base; };
	std::vector<peer_mapping> peer_mappings;

	struct clip_slab
	{
		struct piece { size_t offset, size; bool live; };
		uint8_t* base = nullptr;
		size_t capacity = 0;
		size_t used = 0;
		uint32_t live = 0;
		std::vector<piece> pieces;		// in address order
	};
	std::vector<clip_slab> slabs;
};

namespace
{
	// Environment knobs, all of them, go through these two (INTEGRATION.md 9 lists them).
	//   path_knob: selects between code paths that are EACH held to the oracle bit for bit by tests/ (the other kernel family, the other
	//              key read width, clip slabs off, the three launch ordering, ...) and the test aids of the ordering barrier. Harmless in a
	//              shipped library: whatever they are set to, results are the reference's.
	//   lab_knob:  measurement knobs (LDS padding, occupancy, re-order thresholds) and the one that can break bit exactness
	//              (ACLHIP_SHORT_EXACT_MATH=1). A default build does not read them at all; -DACLHIP_LAB_KNOBS (libaclhip_lab.so,
	//              acl_amd/build.py; implied by -DACLHIP_EXPERIMENTS) does.
	inline const char* path_knob(const char* name) { return std::getenv(name); }
	inline const char* lab_knob(const char* name)
	{
#if defined(ACLHIP_LAB_KNOBS) || defined(ACLHIP_EXPERIMENTS)
		return std::getenv(name);
#else
		(void)name;
		return nullptr;
#endif
	}

	constexpr size_t k_slab_bytes = size_t(32) << 20;
	constexpr size_t k_slab_alignment = 256;

	// nullptr: out of device memory
	uint8_t* allocate_clip_memory(aclhip_context* context, size_t bytes)
	{
		bytes = (bytes + k_slab_alignment - 1) & ~(k_slab_alignment - 1);
		static const bool use_slabs = []() { const char* value = path_knob("ACLHIP_CLIP_SLABS"); return value == nullptr || value[0] != '0'; }();
		if (!use_slabs)
		{
			aclhip_context::clip_slab slab;
			slab.capacity = bytes;
			if (hipMalloc(reinterpret_cast<void**>(&slab.base), slab.capacity) != hipSuccess)
				return nullptr;
			slab.used = bytes;
			slab.live = 1;
			slab.pieces.push_back({ 0, bytes, true });
			context->slabs.push_back(slab);
			return slab.base;
		}
		if (bytes <= k_slab_bytes / 2)
		{
			for (size_t i = context->slabs.size(); i-- > 0;)
			{
				aclhip_context::clip_slab& slab = context->slabs[i];
				if (slab.capacity == k_slab_bytes && slab.capacity - slab.used >= bytes)
				{
					uint8_t* memory = slab.base + slab.used;
					slab.pieces.push_back({ slab.used, bytes, true });
					slab.used += bytes;
					slab.live++;
					return memory;
				}
			}
		}

		// a new slab; clips larger than half a slab get one of their own size
		aclhip_context::clip_slab slab;
		slab.capacity = bytes <= k_slab_bytes / 2 ? k_slab_bytes : bytes;
		if (hipMalloc(reinterpret_cast<void**>(&slab.base), slab.capacity) != hipSuccess)
			return nullptr;
		slab.used = bytes;
		slab.live = 1;
		slab.pieces.push_back({ 0, bytes, true });
		context->slabs.push_back(slab);
		return slab.base;
	}

	// Launch sizing follows the LIVE clips: when the clip that set a maximum goes away the maxima are taken again
	void recompute_launch_maxima(aclhip_context* context)
	{
		context->max_pose_quads = context->max_hierarchy_words = context->max_scalar_tracks = context->max_scalar_frame_bytes = 0;
		context->max_window_animated = context->max_window_key_bytes = 0;
		for (const host_clip& clip : context->clips)
		{
			if (!clip.in_use)
				continue;
			context->max_pose_quads = std::max(context->max_pose_quads, clip.pose_quads);
			context->max_window_animated = std::max(context->max_window_animated, clip.window_animated);
			context->max_window_key_bytes = std::max(context->max_window_key_bytes, clip.window_key_bytes);
			context->max_hierarchy_words = std::max(context->max_hierarchy_words, clip.hierarchy_words);
			context->max_scalar_tracks = std::max(context->max_scalar_tracks, clip.scalar_tracks);
			context->max_scalar_frame_bytes = std::max(context->max_scalar_frame_bytes, clip.scalar_frame_bytes);
		}
	}

	void free_clip_memory(aclhip_context* context, void* memory);

	void release_hierarchy(aclhip_context* context, const uint32_t* d_image)
	{
		for (size_t i = 0; i < context->hierarchies.size(); ++i)
		{
			if (context->hierarchies[i].d_image != d_image)
				continue;
			if (--context->hierarchies[i].num_users == 0)
			{
				free_clip_memory(context, context->hierarchies[i].d_image);		// (a piece of a clip slab: no hipFree, which would synchronize the device)
				context->hierarchies.erase(context->hierarchies.begin() + ptrdiff_t(i));
			}
			return;
		}
	}

	void free_clip_memory(aclhip_context* context, void* memory)
	{
		if (memory == nullptr)
			return;
		const uint8_t* address = static_cast<const uint8_t*>(memory);
		for (size_t i = 0; i < context->slabs.size(); ++i)
		{
			aclhip_context::clip_slab& slab = context->slabs[i];
			if (address < slab.base || address >= slab.base + slab.capacity)
				continue;
			for (aclhip_context::clip_slab::piece& piece : slab.pieces)
				if (slab.base + piece.offset == address)
					piece.live = false;
			while (!slab.pieces.empty() && !slab.pieces.back().live)
			{
				slab.used = slab.pieces.back().offset;
				slab.pieces.pop_back();
			}
			if (--slab.live != 0)
				return;
			// empty: keep one shared slab around for the next registrations, give the rest back
			bool another_empty = slab.capacity != k_slab_bytes;
			for (size_t j = 0; j < context->slabs.size() && !another_empty; ++j)
				another_empty = j != i && context->slabs[j].capacity == k_slab_bytes && context->slabs[j].live == 0;
			if (another_empty)
			{
				(void)hipFree(slab.base);
				context->slabs.erase(context->slabs.begin() + ptrdiff_t(i));
			}
			else
				slab.used = 0;
			return;
		}
	}
}

namespace
{
	constexpr uint32_t k_reserved_table_entries = 1u << 22;		// 4 M clips = 512 MiB of address space, backed on demand
	constexpr uint32_t k_fixed_table_entries = 1u << 18;		// without virtual memory management: 32 MiB, allocated once

	// Streams the context enqueued work on: what a retired clip must outlive. (Callers that destroy a stream they decoded on: an event
	// record on it fails from then on and the stream is dropped from the list.)
	void note_launch_stream(aclhip_context* context, hipStream_t stream)
	{
		// (callers hold context->mutex, shared or exclusive; readers of the list hold it exclusively: no launch is inside this function then)
		std::lock_guard<std::mutex> lock(context->streams_mutex);
		for (hipStream_t known : context->launch_streams)
			if (known == stream)
				return;
		context->launch_streams.push_back(stream);
	}

	hipEvent_t take_event(aclhip_context* context)
	{
		if (!context->event_pool.empty())
		{
			hipEvent_t event = context->event_pool.back();
			context->event_pool.pop_back();
			return event;
		}
		hipEvent_t event = nullptr;
		if (hipEventCreateWithFlags(&event, hipEventDisableTiming) != hipSuccess)
			return nullptr;
		return event;
	}

	// Records "everything enqueued so far" on every launch stream into the item and queues it; nothing is freed here.
	// `record_to_clear` (a clip's record in the device table, or null): cleared on the context's retire stream BEHIND those same points --
	// launches already enqueued still find the clip (kernels read the record when they execute, not when they are enqueued), launches
	// that execute later are refused -- and the item is not recycled before the clear has happened. Nobody waits on the host.
	void retire(aclhip_context* context, aclhip_context::retired_item&& item, device_clip* record_to_clear = nullptr)
	{
		for (size_t i = 0; i < context->launch_streams.size();)
		{
			hipEvent_t event = take_event(context);
			if (event != nullptr && hipEventRecord(event, context->launch_streams[i]) == hipSuccess)
			{
				item.events.push_back(event);
				++i;
				continue;
			}
			// the caller destroyed this stream: its work is over
			(void)hipGetLastError();
			if (event != nullptr)
				context->event_pool.push_back(event);
			context->launch_streams.erase(context->launch_streams.begin() + ptrdiff_t(i));
		}
		if (record_to_clear != nullptr)
		{
			bool ordered = context->retire_stream != nullptr;
			for (size_t i = 0; ordered && i < item.events.size(); ++i)
				ordered = hipStreamWaitEvent(context->retire_stream, item.events[i], 0) == hipSuccess;
			hipEvent_t cleared = ordered ? take_event(context) : nullptr;
			ordered = ordered && cleared != nullptr
				&& hipMemsetAsync(record_to_clear, 0, sizeof(device_clip), context->retire_stream) == hipSuccess
				&& hipEventRecord(cleared, context->retire_stream) == hipSuccess;
			if (ordered)
				item.events.push_back(cleared);
			else
			{
				// (no stream ordered clear possible: fall back to what round 2 did, a clear the host waits for)
				(void)hipGetLastError();
				if (cleared != nullptr)
					context->event_pool.push_back(cleared);
				(void)hipMemsetAsync(record_to_clear, 0, sizeof(device_clip), context->copy_stream);
				(void)hipStreamSynchronize(context->copy_stream);
			}
		}
		context->retired.push_back(std::move(item));
	}

	void release_hierarchy(aclhip_context* context, const uint32_t* d_image);
	void free_clip_memory(aclhip_context* context, void* memory);

	// Frees what retired items held once their events have completed; `wait` = block for them (aclhip_destroy)
	void collect_retired(aclhip_context* context, bool wait)
	{
		for (size_t i = 0; i < context->retired.size();)
		{
			aclhip_context::retired_item& item = context->retired[i];
			bool done = true;
			for (hipEvent_t event : item.events)
			{
				const hipError_t status = wait ? hipEventSynchronize(event) : hipEventQuery(event);
				if (status == hipErrorNotReady)
				{
					done = false;
					break;
				}
				if (status != hipSuccess)
					(void)hipGetLastError();		// a stream that died with its event: nothing left to wait for
			}
			if (!done)
			{
				++i;
				continue;
			}
			for (hipEvent_t event : item.events)
				context->event_pool.push_back(event);
			if (item.clip_memory != nullptr)
				free_clip_memory(context, item.clip_memory);
			if (item.hierarchy != nullptr)
				release_hierarchy(context, item.hierarchy);
			if (item.slot != ACLHIP_INVALID_HANDLE)
				context->free_slots.push_back(item.slot);
			for (uint8_t* memory : item.database_memory)
				if (memory != nullptr)
					(void)hipFree(memory);
			for (uint8_t* memory : item.database_pinned)
				if (memory != nullptr)
					(void)hipHostFree(memory);
			if (item.device_memory != nullptr)
				(void)hipFree(item.device_memory);
			context->deferred_frees_completed++;
			context->retired[i] = std::move(context->retired.back());
			context->retired.pop_back();
		}
	}

	// Host bytes -> device memory without touching the caller's streams: staged in pinned memory, copied on the context's copy
	// stream. The copies of one call are waited for together (finish_uploads): only the calling thread blocks.
	bool stage_upload(aclhip_context* context, void* device_destination, const void* host_source, size_t bytes, size_t& staging_used)
	{
		if (bytes == 0)
			return true;
		const size_t needed = staging_used + ((bytes + 255) & ~size_t(255));
		if (needed > context->pinned_staging_bytes)
		{
			// grow: earlier pieces of this call are still being copied from the old buffer
			if (staging_used != 0 && hipStreamSynchronize(context->copy_stream) != hipSuccess)
				return false;
			staging_used = 0;
			const size_t capacity = std::max<size_t>((bytes + (size_t(1) << 20)) & ~((size_t(1) << 20) - 1), size_t(4) << 20);
			if (capacity > context->pinned_staging_bytes)
			{
				if (context->pinned_staging != nullptr)
					(void)hipHostFree(context->pinned_staging);
				context->pinned_staging = nullptr;
				context->pinned_staging_bytes = 0;
				if (hipHostMalloc(reinterpret_cast<void**>(&context->pinned_staging), capacity, hipHostMallocDefault) != hipSuccess)
					return false;
				context->pinned_staging_bytes = capacity;
			}
		}
		std::memcpy(context->pinned_staging + staging_used, host_source, bytes);
		if (hipMemcpyAsync(device_destination, context->pinned_staging + staging_used, bytes, hipMemcpyHostToDevice, context->copy_stream) != hipSuccess)
			return false;
		staging_used += (bytes + 255) & ~size_t(255);
		return true;
	}

	bool finish_uploads(aclhip_context* context)
	{
		return hipStreamSynchronize(context->copy_stream) == hipSuccess;
	}

	// Makes `stream` wait for everything enqueued so far on the streams this context has launched on (events, no host wait)
	bool order_stream_behind_launches(aclhip_context* context, hipStream_t stream)
	{
		for (size_t i = 0; i < context->launch_streams.size();)
		{
			hipEvent_t event = take_event(context);
			if (event != nullptr && hipEventRecord(event, context->launch_streams[i]) == hipSuccess)
			{
				const bool waiting = hipStreamWaitEvent(stream, event, 0) == hipSuccess;
				context->event_pool.push_back(event);		// (the wait holds its own reference to the recorded state)
				if (!waiting)
					return false;
				++i;
				continue;
			}
			(void)hipGetLastError();		// the caller destroyed this stream: its work is over
			if (event != nullptr)
				context->event_pool.push_back(event);
			context->launch_streams.erase(context->launch_streams.begin() + ptrdiff_t(i));
		}
		return true;
	}
}

namespace
{
	thread_local std::string t_last_error;

	aclhip_status fail(const aclhip_context* context, aclhip_status status, const char* format, ...)
	{
		char buffer[512];
		va_list args;
		va_start(args, format);
		std::vsnprintf(buffer, sizeof(buffer), format, args);
		va_end(args);
		(void)context;
		t_last_error = buffer;		// per thread: contexts are shared between threads, messages are not
		return status;
	}

	#define ACLHIP_CHECK_HIP(context, expression) \
		do { const hipError_t hip_status_ = (expression); if (hip_status_ != hipSuccess) return fail((context), ACLHIP_ERROR_DEVICE, "%s failed: %s", #expression, hipGetErrorString(hip_status_)); } while (0)

	// Makes the context's device current for the duration of a call; a no-op (one thread-local read) when it already is,
	// which is the one-process-per-GPU case the launch path cares about.
	struct device_guard
	{
		int previous = -1;
		bool switched = false;
		bool ok = false;
		explicit device_guard(int device)
		{
			if (hipGetDevice(&previous) != hipSuccess)
				return;
			if (previous == device)
				ok = true;
			else
			{
				ok = hipSetDevice(device) == hipSuccess;
				switched = ok;
			}
		}
		~device_guard() { if (switched) (void)hipSetDevice(previous); }
	};

	// compressed_tracks::is_valid (core/impl/compressed_tracks.impl.h:278-301) + bounds checks so that a decode can never read outside the blob
	// Scalar track lists: every offset of the scalar_tracks_header and the whole animated stream must lie inside the buffer
	// (compressed_tracks::is_valid only checks tag / version / hash, core/impl/compressed_tracks.impl.h:278-301; the device reads
	// through these offsets, so they are checked here).
	aclhip_status validate_scalar_clip(const aclhip_context* context, const uint8_t* blob)
	{
		const raw_buffer_header& buffer_header = *reinterpret_cast<const raw_buffer_header*>(blob);
		const tracks_header& header = *reinterpret_cast<const tracks_header*>(blob + k_tracks_header_offset);
		if (header.has_database())
			return fail(context, ACLHIP_ERROR_INVALID_CLIP, "database decompression is not supported for scalar tracks");	// decompression.scalar.h:107-108
		if (header.num_tracks == 0 || header.num_samples == 0)
			return ACLHIP_OK;
		if (!(header.sample_rate > 0.0f))
			return fail(context, ACLHIP_ERROR_INVALID_CLIP, "Invalid sample rate");
		if (buffer_header.size < k_transform_header_offset + sizeof(scalar_tracks_header))
			return fail(context, ACLHIP_ERROR_INVALID_CLIP, "Invalid size");

		const scalar_tracks_header& sh = *reinterpret_cast<const scalar_tracks_header*>(blob + k_transform_header_offset);
		const uint64_t limit = buffer_header.size - k_transform_header_offset;
		const uint32_t num_components = scalar_track_num_components(header.track_type);
		const uint8_t* num_bits_at_bit_rate = header.version == k_version_first ? k_bit_rate_num_bits_v0 : k_bit_rate_num_bits;
		const uint32_t num_bit_rates = header.version == k_version_first ? sizeof(k_bit_rate_num_bits_v0) : sizeof(k_bit_rate_num_bits);
		if (uint64_t(sh.metadata_per_track) + header.num_tracks > limit)
			return fail(context, ACLHIP_ERROR_INVALID_CLIP, "Track metadata points outside of the buffer");

		const uint8_t* bit_rates = reinterpret_cast<const uint8_t*>(&sh) + sh.metadata_per_track;
		uint64_t num_constant = 0, num_ranged = 0, bits_per_frame = 0;
		for (uint32_t track = 0; track < header.num_tracks; ++track)
		{
			if (bit_rates[track] >= num_bit_rates)
				return fail(context, ACLHIP_ERROR_INVALID_CLIP, "Invalid bit rate: %u", uint32_t(bit_rates[track]));
			const uint32_t num_bits = num_bits_at_bit_rate[bit_rates[track]];
			num_constant += num_bits == 0 ? 1 : 0;
			num_ranged += (num_bits != 0 && num_bits != 32) ? 1 : 0;
			bits_per_frame += uint64_t(num_bits) * num_components;
		}
		if (bits_per_frame != sh.num_bits_per_frame || bits_per_frame > k_quad_ordinal_mask)
			return fail(context, ACLHIP_ERROR_INVALID_CLIP, "Track bit rates add up to %llu bits per frame, header says %u", static_cast<unsigned long long>(bits_per_frame), sh.num_bits_per_frame);
		if (((sh.track_constant_values | sh.track_range_values) & 3u) != 0)
			return fail(context, ACLHIP_ERROR_INVALID_CLIP, "Header offsets are not 4 byte aligned");
		if (uint64_t(sh.track_constant_values) + num_constant * num_components * 4 > limit
			|| uint64_t(sh.track_range_values) + num_ranged * num_components * 8 > limit
			|| uint64_t(sh.track_animated_values) + (bits_per_frame * header.num_samples + 7) / 8 > limit
			|| (bits_per_frame * header.num_samples) >> 32 != 0)
			return fail(context, ACLHIP_ERROR_INVALID_CLIP, "Header offsets point outside of the buffer");
		return ACLHIP_OK;
	}

	aclhip_status validate_clip(const aclhip_context* context, const uint8_t* blob, uint64_t size, int check_hash)
	{
		if (blob == nullptr || size < k_transform_header_offset)
			return fail(context, ACLHIP_ERROR_INVALID_CLIP, "buffer is not a valid compressed_tracks instance (too small)");

		const raw_buffer_header& buffer_header = *reinterpret_cast<const raw_buffer_header*>(blob);
		const tracks_header& header = *reinterpret_cast<const tracks_header*>(blob + k_tracks_header_offset);
		if (header.tag != k_tag_compressed_tracks)
			return fail(context, ACLHIP_ERROR_INVALID_CLIP, "Invalid tag");
		if (header.algorithm_type != k_algorithm_uniformly_sampled)
			return fail(context, ACLHIP_ERROR_INVALID_CLIP, "Invalid algorithm type");
		if (header.version < k_version_first || header.version > k_version_latest)
			return fail(context, ACLHIP_ERROR_INVALID_CLIP, "Invalid algorithm version");
		const bool is_scalar_list = scalar_track_num_components(header.track_type) != 0;
		if (buffer_header.size > size || buffer_header.size < k_transform_header_offset + (is_scalar_list ? 0 : sizeof(transform_tracks_header)))
			return fail(context, ACLHIP_ERROR_INVALID_CLIP, "Invalid size");
		if (check_hash && hash32(blob + sizeof(raw_buffer_header), buffer_header.size - sizeof(raw_buffer_header)) != buffer_header.hash)
			return fail(context, ACLHIP_ERROR_INVALID_CLIP, "Invalid hash");

		if (scalar_track_num_components(header.track_type) != 0)
			return validate_scalar_clip(context, blob);
		if (header.track_type != k_track_type_qvvf)
			return fail(context, ACLHIP_ERROR_UNSUPPORTED_FORMAT, "unsupported track type %u", uint32_t(header.track_type));
		if (header.num_tracks == 0)
			return ACLHIP_OK;
		// every packed format the reference's decoder takes (debug_transform_decompression_settings, decompression_settings.h:236-262):
		// quatf_full, quatf_drop_w_full, quatf_drop_w_variable; vector3f_full, vector3f_variable (one header bit each: always one of the two)
		if (header.rotation_format() != k_rotation_quatf_full && header.rotation_format() != k_rotation_quatf_drop_w_full && header.rotation_format() != k_rotation_quatf_drop_w_variable)
			return fail(context, ACLHIP_ERROR_UNSUPPORTED_FORMAT, "unknown rotation format %u", uint32_t(header.rotation_format()));
		if (header.num_samples == 0 || !(header.sample_rate > 0.0f))
			return fail(context, ACLHIP_ERROR_INVALID_CLIP, "Invalid sample count or rate");

		const transform_tracks_header& th = *reinterpret_cast<const transform_tracks_header*>(blob + k_transform_header_offset);
		const uint64_t blob_size = buffer_header.size;
		const uint64_t tbase = k_transform_header_offset;
		const bool stripped = header.has_stripped_keyframes() || header.has_database();
		const uint32_t segment_header_size = stripped ? sizeof(stripped_segment_header) : sizeof(segment_header);
		// (64 bit: num_tracks is untrusted, and 0xFFFFFFFF + 15 wraps to 14 -- no sub-track type words at all, the bounds test below passed,
		// and registration went on to size its tables for 4 G tracks: found with the validators under AddressSanitizer, round 5)
		const uint64_t num_entries = (uint64_t(header.num_tracks) + 15) / 16;
		// Only the VARIABLE formats have per sub-track metadata (a format byte and, in multi segment clips, six range bytes per segment;
		// a clip range entry): animated_track_cache.transform.h:1254-1300, write_stream_data.h:158-197, write_range_data.h:79-207
		const bool rotations_variable = header.rotation_format() == k_rotation_quatf_drop_w_variable;
		const bool translations_variable = header.translation_format() == k_vector_vector3f_variable;
		const bool scales_variable = header.scale_format() == k_vector_vector3f_variable;
		const uint64_t num_rotations_padded = rotations_variable ? ((uint64_t(th.num_animated_rotation_sub_tracks) + 3) & ~uint64_t(3)) : 0;
		const uint64_t num_variable_translations = translations_variable ? th.num_animated_translation_sub_tracks : 0;
		const uint64_t num_variable_scales = scales_variable ? th.num_animated_scale_sub_tracks : 0;
		const uint64_t constant_rotation_size = header.rotation_format() == k_rotation_quatf_full ? 16 : 12;		// constant_track_cache.transform.h:102-110

		if (th.num_segments == 0)
			return fail(context, ACLHIP_ERROR_INVALID_CLIP, "Invalid segment count");
		// (registration keeps 16 bytes per sample: a blob without animated sub-tracks could otherwise name any count -- 2^24 samples are 155 hours at 30 Hz)
		if (header.num_samples > (1u << 24))
			return fail(context, ACLHIP_ERROR_UNSUPPORTED_FORMAT, "%u samples: clips of more than 16 777 216 samples are not supported", header.num_samples);
		// A segment holds at most 32 samples (the compressor cuts at 31, segment_context.h:55-66; stripped keyframes and database tiers name a
		// segment's keyframes in 32 bits) -- unless EVERY format is a full one: such a clip is never segmented and is one segment of any length
		// (compress.transform.impl.h:168-176). (Round 6 first lifted the limit for every clip without stripped keyframes: a multi segment clip
		// whose segment count was edited to 1 then registered, and the reference, its restatement and these kernels each read the bytes behind
		// segment 0 their own way -- three poses for one blob, found by tools/fuzz_gpu_mutated.py.)
		const bool segments_of_any_length = !stripped && !rotations_variable && !translations_variable && (!header.has_scale() || !scales_variable);
		if (!segments_of_any_length && uint64_t(header.num_samples) > uint64_t(th.num_segments) * 32)
			return fail(context, ACLHIP_ERROR_INVALID_CLIP, "%u samples cannot fit in %u segments of at most 32", header.num_samples, th.num_segments);
		if (uint64_t(th.num_animated_variable_sub_tracks) != num_rotations_padded + num_variable_translations + num_variable_scales)
			return fail(context, ACLHIP_ERROR_INVALID_CLIP, "Inconsistent animated sub-track counts");
		// (a track has one sub-track of each kind: no count exceeds the track count. The sum above bounded the three counts while every
		// format was variable; a full format's count appears in no sum -- 0xFFFFFFFF animated rotations + 2 translations wrapped to ONE
		// table entry at registration: found by the mutated-clip fuzz under AddressSanitizer, round 6)
		if (th.num_animated_rotation_sub_tracks > header.num_tracks || th.num_animated_translation_sub_tracks > header.num_tracks || th.num_animated_scale_sub_tracks > header.num_tracks
			|| th.num_constant_rotation_samples > header.num_tracks || th.num_constant_translation_samples > header.num_tracks || th.num_constant_scale_samples > header.num_tracks)
			return fail(context, ACLHIP_ERROR_INVALID_CLIP, "More sub-tracks of a kind than tracks");
		if (tbase + th.segment_headers_offset + uint64_t(segment_header_size) * th.num_segments > blob_size
			|| tbase + th.sub_track_types_offset + num_entries * 4 * (header.has_scale() ? 3 : 2) > blob_size
			|| tbase + th.constant_track_data_offset + constant_rotation_size * th.num_constant_rotation_samples + 12ull * (uint64_t(th.num_constant_translation_samples) + th.num_constant_scale_samples) > blob_size
			|| tbase + th.clip_range_data_offset + 24ull * ((rotations_variable ? uint64_t(th.num_animated_rotation_sub_tracks) : 0) + num_variable_translations + num_variable_scales) > blob_size)
			return fail(context, ACLHIP_ERROR_INVALID_CLIP, "Header offsets point outside of the buffer");
		// (the reference's writer aligns every one of these sections to 4 bytes, and the host reads them as words and floats)
		if (((th.segment_headers_offset | th.sub_track_types_offset | th.constant_track_data_offset | th.clip_range_data_offset) & 3u) != 0
			|| (header.has_database() && (th.database_header_offset & 3u) != 0))
			return fail(context, ACLHIP_ERROR_INVALID_CLIP, "Header offsets are not 4 byte aligned");
		if (th.num_segments > 1 && tbase + k_segment_start_indices_offset + 4ull * (th.num_segments + 1) > blob_size)
			return fail(context, ACLHIP_ERROR_INVALID_CLIP, "Segment start indices point outside of the buffer");
		if (header.has_database() && tbase + th.database_header_offset + sizeof(tracks_database_header) > blob_size)
			return fail(context, ACLHIP_ERROR_INVALID_CLIP, "Database header points outside of the buffer");

		const uint32_t* segment_start_indices = th.num_segments > 1 ? reinterpret_cast<const uint32_t*>(blob + tbase + k_segment_start_indices_offset) : nullptr;
		// (the list ends with 0xFFFFFFFF: what stops the reference's scan for a key's segment, decompression.transform.h:374-409 -- the
		// sample records here do not need it, a reference decoder fed the same blob would walk into a segment that is not there)
		if (segment_start_indices != nullptr && segment_start_indices[th.num_segments] != 0xFFFFFFFFu)
			return fail(context, ACLHIP_ERROR_INVALID_CLIP, "The segment start indices do not end with 0xFFFFFFFF");
		for (uint32_t i = 0; i < th.num_segments; ++i)
		{
			const segment_header& sh = *reinterpret_cast<const segment_header*>(blob + tbase + th.segment_headers_offset + size_t(i) * segment_header_size);
			// transform_tracks_header::get_segment_data (core/impl/compressed_headers.h:309-324) in 64 bit arithmetic: segment_data is
			// an untrusted 32 bit offset, a value near 2^32 must not wrap into the buffer
			const uint64_t format_offset = tbase + uint64_t(sh.segment_data);
			const uint64_t range_offset = (format_offset + th.num_animated_variable_sub_tracks + 1u) & ~uint64_t(1);
			const uint64_t animated_offset = (range_offset + (th.num_segments > 1 ? 6ull * th.num_animated_variable_sub_tracks : 0ull) + 3u) & ~uint64_t(3);
			if (sh.segment_data == 0xFFFFFFFFu || animated_offset > blob_size || sh.animated_rotation_bit_size > sh.animated_pose_bit_size)
				return fail(context, ACLHIP_ERROR_INVALID_CLIP, "Segment %u points outside of the buffer", i);

			// every keyframe the clip itself stores (all of them, or the ones its sample_indices keep when keyframes were stripped or
			// moved to a database) must lie inside the buffer
			const uint64_t start = th.num_segments > 1 ? segment_start_indices[i] : 0;
			const uint64_t end = th.num_segments > 1 && i + 1 < th.num_segments ? segment_start_indices[i + 1] : header.num_samples;
			if (start >= end || end > header.num_samples || (!segments_of_any_length && end - start > 32))
				return fail(context, ACLHIP_ERROR_INVALID_CLIP, "segment %u has an invalid sample range [%llu, %llu)", i, static_cast<unsigned long long>(start), static_cast<unsigned long long>(end));
			if (stripped)
			{
				// A segment's first and last sample are always among the keyframes it keeps (the compressor strips and moves to a database
				// only what lies between them, compress.transform.impl.h / compress.database.impl.h), and it keeps none past its last
				// sample: the seek's "nearest keyframe that is present" (decompression.transform.h:272-362) has no answer otherwise -- a
				// count of leading zeros of zero, which the reference's CPU and this device define differently.
				const uint32_t sample_indices = reinterpret_cast<const stripped_segment_header&>(sh).sample_indices;
				const uint32_t count = uint32_t(end - start);		// 1 .. 32
				const uint32_t past_the_end = count < 32 ? (0xFFFFFFFFu >> count) : 0u;
				if ((sample_indices & 0x80000000u) == 0 || (sample_indices & (0x80000000u >> (count - 1))) == 0 || (sample_indices & past_the_end) != 0)
					return fail(context, ACLHIP_ERROR_INVALID_CLIP, "segment %u: the keyframes it keeps (%08x) do not span its %u samples", i, sample_indices, count);
			}
			const uint64_t stored = stripped ? uint64_t(__builtin_popcount(reinterpret_cast<const stripped_segment_header&>(sh).sample_indices)) : end - start;
			const uint64_t animated_end = animated_offset + (uint64_t(sh.animated_pose_bit_size) * stored + 7) / 8;
			// (a keyframe's bit offset inside its segment is a 32 bit product, here and in the reference: decompression.transform.h:533-534)
			if (animated_end > blob_size || uint64_t(sh.animated_pose_bit_size) * stored > 0xFFFFFFFFull)
				return fail(context, ACLHIP_ERROR_INVALID_CLIP, "segment %u animated data points outside of the buffer", i);
			// ... and in front of the next segment's data: the writer lays the segments out one behind the other
			// (compression/impl/write_segment_data.h), so a segment whose sample range says it stores more keyframes than lie between its
			// data and the next segment's claims keyframes that are not there. Such "keyframes" are the next segment's format bytes; the
			// reference (which unpacks four sub-tracks at a time and reads past them), its restatement and these kernels each make something
			// else of them -- three different poses for one blob, measured. Refused.
			if (i + 1 < th.num_segments)
			{
				const segment_header& next = *reinterpret_cast<const segment_header*>(blob + tbase + th.segment_headers_offset + size_t(i + 1) * segment_header_size);
				if (next.segment_data != 0xFFFFFFFFu && animated_end > tbase + uint64_t(next.segment_data))
					return fail(context, ACLHIP_ERROR_INVALID_CLIP, "segment %u claims keyframes that overlap the data of segment %u", i, i + 1);
			}
		}

		return ACLHIP_OK;
	}

	uint32_t sub_track_class(const uint32_t* types, uint32_t track_index)
	{
		return (types[track_index / 16] >> ((15 - (track_index % 16)) * 2)) & 3u;
	}

	// The clip table never moves (launches in flight and captured hipGraphs hold its address): records are added by backing more of
	// the reserved address range. Called under the registry lock.
	aclhip_status grow_clip_table(aclhip_context* context, uint32_t needed)
	{
		if (needed <= context->d_clips_capacity)
			return ACLHIP_OK;

		if (context->d_clips == nullptr)
		{
			// first call (aclhip_create): reserve the range, or fall back to one fixed allocation
			hipMemAllocationProp properties = {};
			properties.type = hipMemAllocationTypePinned;
			properties.location.type = hipMemLocationTypeDevice;
			properties.location.id = context->device;
			size_t granularity = 0;
			void* range = nullptr;
			static const bool allow_virtual = []() { const char* value = path_knob("ACLHIP_VIRTUAL_CLIP_TABLE"); return value == nullptr || value[0] != '0'; }();
			if (allow_virtual && hipMemGetAllocationGranularity(&granularity, &properties, hipMemAllocationGranularityRecommended) == hipSuccess && granularity != 0
				&& hipMemAddressReserve(&range, size_t(k_reserved_table_entries) * sizeof(device_clip), granularity, nullptr, 0) == hipSuccess && range != nullptr)
			{
				context->table_is_virtual = true;
				context->table_granularity = granularity;
				context->table_reserved_bytes = size_t(k_reserved_table_entries) * sizeof(device_clip);
				context->d_clips = static_cast<device_clip*>(range);
			}
			else
			{
				(void)hipGetLastError();
				device_clip* table = nullptr;
				ACLHIP_CHECK_HIP(context, hipMalloc(reinterpret_cast<void**>(&table), sizeof(device_clip) * k_fixed_table_entries));
				ACLHIP_CHECK_HIP(context, hipMemsetAsync(table, 0, sizeof(device_clip) * k_fixed_table_entries, context->copy_stream));
				ACLHIP_CHECK_HIP(context, hipStreamSynchronize(context->copy_stream));
				context->d_clips = table;
				context->d_clips_capacity = k_fixed_table_entries;
				return needed <= k_fixed_table_entries ? ACLHIP_OK : fail(context, ACLHIP_ERROR_OUT_OF_MEMORY, "the clip table holds %u clips", k_fixed_table_entries);
			}
		}

		if (!context->table_is_virtual)
			return fail(context, ACLHIP_ERROR_OUT_OF_MEMORY, "the clip table holds %u clips", context->d_clips_capacity);
		if (size_t(needed) * sizeof(device_clip) > context->table_reserved_bytes)
			return fail(context, ACLHIP_ERROR_OUT_OF_MEMORY, "the clip table holds %u clips", k_reserved_table_entries);

		// back [mapped, mapped + step): at least 16384 records (2 MiB) at a time
		const size_t wanted = std::max<size_t>(size_t(needed) * sizeof(device_clip), context->table_mapped_bytes + (size_t(16384) * sizeof(device_clip)));
		const size_t new_mapped = std::min((wanted + context->table_granularity - 1) / context->table_granularity * context->table_granularity, context->table_reserved_bytes);
		const size_t step = new_mapped - context->table_mapped_bytes;
		hipMemAllocationProp properties = {};
		properties.type = hipMemAllocationTypePinned;
		properties.location.type = hipMemLocationTypeDevice;
		properties.location.id = context->device;
		hipMemGenericAllocationHandle_t handle;
		ACLHIP_CHECK_HIP(context, hipMemCreate(&handle, step, &properties, 0));
		uint8_t* address = reinterpret_cast<uint8_t*>(context->d_clips) + context->table_mapped_bytes;
		hipMemAccessDesc access = {};
		access.location.type = hipMemLocationTypeDevice;
		access.location.id = context->device;
		access.flags = hipMemAccessFlagsProtReadWrite;
		if (hipMemMap(address, step, 0, handle, 0) != hipSuccess || hipMemSetAccess(address, step, &access, 1) != hipSuccess)
		{
			(void)hipMemRelease(handle);
			return fail(context, ACLHIP_ERROR_OUT_OF_MEMORY, "backing %zu more bytes of the clip table failed", step);
		}
		context->table_handles.push_back(handle);
		ACLHIP_CHECK_HIP(context, hipMemsetAsync(address, 0, step, context->copy_stream));
		ACLHIP_CHECK_HIP(context, hipStreamSynchronize(context->copy_stream));
		context->table_mapped_bytes = new_mapped;
		context->d_clips_capacity = uint32_t(new_mapped / sizeof(device_clip));
		return ACLHIP_OK;
	}

	aclhip_status resolve_params(const aclhip_context* context, const aclhip_decompress_params* params, decode_params& out)
	{
		aclhip_decompress_params defaults;
		if (params == nullptr)
		{
			aclhip_default_params(&defaults);
			params = &defaults;
		}

		if (params->rounding_policy > ACLHIP_ROUND_PER_TRACK || params->looping_policy > ACLHIP_LOOP_AS_COMPRESSED || params->normalization > ACLHIP_NORMALIZE_ALWAYS)
			return fail(context, ACLHIP_ERROR_INVALID_ARGUMENT, "invalid rounding / looping / normalization policy");
		const auto valid_mode = [](uint32_t mode, bool scale) { return mode <= ACLHIP_DEFAULT_VARIABLE || mode == ACLHIP_DEFAULT_BIND_POSE || (scale && mode == ACLHIP_DEFAULT_LEGACY); };
		if (!valid_mode(params->default_rotation_mode, false) || !valid_mode(params->default_translation_mode, false) || !valid_mode(params->default_scale_mode, true))
			return fail(context, ACLHIP_ERROR_INVALID_ARGUMENT, "invalid default sub-track mode (legacy is only valid for scale)");
		if (params->rounding_policy == ACLHIP_ROUND_PER_TRACK && params->per_track_rounding == 0)
			return fail(context, ACLHIP_ERROR_INVALID_ARGUMENT, "sample_rounding_policy::per_track needs per_track_rounding enabled (decompression_settings::is_per_track_rounding_supported)");
		const bool needs_values = params->default_rotation_mode == ACLHIP_DEFAULT_VARIABLE || params->default_translation_mode == ACLHIP_DEFAULT_VARIABLE || params->default_scale_mode == ACLHIP_DEFAULT_VARIABLE;
		if (needs_values && params->default_values == nullptr)
			return fail(context, ACLHIP_ERROR_INVALID_ARGUMENT, "variable default sub-tracks need default_values");

		out.default_values = params->default_values;
		out.track_rounding_policies = params->track_rounding_policies;
		out.instance_rounding_policies = params->instance_rounding_policies;
		out.instance_rows = nullptr;
		out.time_indices = nullptr;
		out.skip_tracks = nullptr;
		out.mask_table = nullptr;
		out.instance_masks = nullptr;
		out.instance_track_counts = nullptr;
		out.instance_looping_policies = params->instance_looping_policies;
		if ((params->track_rounding_table != nullptr) != (params->instance_rounding_tables != nullptr))
			return fail(context, ACLHIP_ERROR_INVALID_ARGUMENT, "track_rounding_table and instance_rounding_tables come together");
		if (params->track_rounding_table != nullptr && params->track_rounding_stride == 0)
			return fail(context, ACLHIP_ERROR_INVALID_ARGUMENT, "track_rounding_stride: the bytes of one table of track_rounding_table (at least the tracks of the largest clip of the batch)");
		out.track_rounding_table = params->track_rounding_table;
		out.instance_rounding_tables = params->instance_rounding_tables;
		out.track_rounding_stride = params->track_rounding_stride;
		out.mask_stride = 0;
		out.layout = ACLHIP_LAYOUT_QVV48;
		out.skip_mask = 0;
		out.items_per_wave = 1;
		out.clips_by_caller_instance = 0;
		if ((params->flags & ~uint32_t(ACLHIP_DECODE_FAST)) != 0)
			return fail(context, ACLHIP_ERROR_INVALID_ARGUMENT, "unknown aclhip_decompress_params::flags %#x", params->flags);
		out.fast_math = (params->flags & ACLHIP_DECODE_FAST) != 0 ? 1 : 0;
		out.user_defaults = (params->default_values != nullptr || params->default_rotation_mode == ACLHIP_DEFAULT_BIND_POSE || params->default_translation_mode == ACLHIP_DEFAULT_BIND_POSE
			|| params->default_scale_mode == ACLHIP_DEFAULT_BIND_POSE) ? 1 : 0;
		out.rounding_policy = params->rounding_policy;
		out.looping_policy = params->looping_policy;
		out.normalization = params->normalization;
		out.per_track_rounding = params->per_track_rounding;
		out.default_modes[0] = params->default_rotation_mode;
		out.default_modes[1] = params->default_translation_mode;
		out.default_modes[2] = params->default_scale_mode;
		// the common case gets a branch-light store loop: track_writer defaults (core/track_writer.h:161-163) without user values,
		// and no re-normalization of constant rotations
		out.standard_default_modes = (params->default_rotation_mode == ACLHIP_DEFAULT_CONSTANT && params->default_translation_mode == ACLHIP_DEFAULT_CONSTANT
			&& params->default_scale_mode == ACLHIP_DEFAULT_LEGACY && params->default_values == nullptr) ? 1 : 0;
		out.standard_defaults = (out.standard_default_modes != 0 && params->normalization != ACLHIP_NORMALIZE_ALWAYS) ? 1 : 0;
		return ACLHIP_OK;
	}
}

extern "C" const char* aclhip_status_string(aclhip_status status)
{
	switch (status)
	{
	case ACLHIP_OK: return "ok";
	case ACLHIP_ERROR_INVALID_ARGUMENT: return "invalid argument";
	case ACLHIP_ERROR_INVALID_CLIP: return "invalid compressed_tracks";
	case ACLHIP_ERROR_UNSUPPORTED_FORMAT: return "unsupported track type or format";
	case ACLHIP_ERROR_UNKNOWN_CLIP: return "unknown clip handle";
	case ACLHIP_ERROR_OUT_OF_MEMORY: return "out of memory";
	case ACLHIP_ERROR_DEVICE: return "HIP error";
	case ACLHIP_ERROR_NO_DEVICE: return "no HIP device";
	case ACLHIP_ERROR_UNKNOWN_DATABASE: return "unknown database handle";
	case ACLHIP_ERROR_NOT_IN_DATABASE: return "clip is not contained in the database";
	case ACLHIP_ERROR_NO_METADATA: return "the clip does not carry that optional metadata";
	}
	return "unknown status";
}

extern "C" const char* aclhip_last_error_message(const aclhip_context* context)
{
	(void)context;
	return t_last_error.c_str();
}

extern "C" void aclhip_default_params(aclhip_decompress_params* out_params)
{
	if (out_params == nullptr)
		return;
	std::memset(out_params, 0, sizeof(*out_params));
	out_params->rounding_policy = ACLHIP_ROUND_NONE;
	out_params->looping_policy = ACLHIP_LOOP_AS_COMPRESSED;
	out_params->normalization = ACLHIP_NORMALIZE_LERP_ONLY;			// default_transform_decompression_settings (decompression_settings.h:227)
	out_params->per_track_rounding = 0;									// decompression_settings.h:231
	out_params->default_rotation_mode = ACLHIP_DEFAULT_CONSTANT;		// core/track_writer.h:161-163
	out_params->default_translation_mode = ACLHIP_DEFAULT_CONSTANT;
	out_params->default_scale_mode = ACLHIP_DEFAULT_LEGACY;
}

extern "C" uint32_t aclhip_abi_version(void)
{
	return ACLHIP_ABI_VERSION;
}

extern "C" aclhip_status aclhip_create(int device_index, aclhip_context** out_context)
{
	if (out_context == nullptr)
		return ACLHIP_ERROR_INVALID_ARGUMENT;
	*out_context = nullptr;

	int device_count = 0;
	if (hipGetDeviceCount(&device_count) != hipSuccess || device_count <= 0)
		return ACLHIP_ERROR_NO_DEVICE;
	if (device_index < 0 || device_index >= device_count)
		return ACLHIP_ERROR_INVALID_ARGUMENT;

	aclhip_context* context = new (std::nothrow) aclhip_context();
	if (context == nullptr)
		return ACLHIP_ERROR_OUT_OF_MEMORY;
	context->device = device_index;
	{
		const char* force_generic = path_knob("ACLHIP_FORCE_GENERIC_KERNEL");
		context->force_generic_kernel = force_generic != nullptr && force_generic[0] == '1';
	}

	device_guard guard(device_index);
	{
		int compute_units = 0;
		if (guard.ok && hipDeviceGetAttribute(&compute_units, hipDeviceAttributeMultiprocessorCount, device_index) == hipSuccess && compute_units > 0)
			context->num_compute_units = uint32_t(compute_units);
	}
	if (!guard.ok || hipStreamCreateWithFlags(&context->copy_stream, hipStreamNonBlocking) != hipSuccess
		|| hipStreamCreateWithFlags(&context->retire_stream, hipStreamNonBlocking) != hipSuccess
		|| hipMalloc(reinterpret_cast<void**>(&context->d_rejected), 2 * sizeof(unsigned long long)) != hipSuccess
		|| hipMemsetAsync(context->d_rejected, 0, 2 * sizeof(unsigned long long), context->copy_stream) != hipSuccess
		|| hipStreamSynchronize(context->copy_stream) != hipSuccess)
	{
		if (context->copy_stream != nullptr)
			(void)hipStreamDestroy(context->copy_stream);
		if (context->retire_stream != nullptr)
			(void)hipStreamDestroy(context->retire_stream);
		delete context;
		return ACLHIP_ERROR_DEVICE;
	}

	const aclhip_status status = grow_clip_table(context, 1);
	if (status != ACLHIP_OK)
	{
		(void)hipFree(context->d_rejected);
		(void)hipStreamDestroy(context->copy_stream);
		(void)hipStreamDestroy(context->retire_stream);
		delete context;
		return status;
	}

	*out_context = context;
	return ACLHIP_OK;
}

void free_instance_lists(aclhip_context* context);		// host_lists.inl

extern "C" void aclhip_destroy(aclhip_context* context)
{
	if (context == nullptr)
		return;
	{
		device_guard guard(context->device);
		(void)hipDeviceSynchronize();
		collect_retired(context, true);
		free_instance_lists(context);
		for (hipEvent_t event : context->event_pool)
			(void)hipEventDestroy(event);
		if (context->pinned_staging != nullptr)
			(void)hipHostFree(context->pinned_staging);
		if (context->copy_stream != nullptr)
			(void)hipStreamDestroy(context->copy_stream);
		if (context->retire_stream != nullptr)
			(void)hipStreamDestroy(context->retire_stream);
		for (aclhip_context::clip_slab& slab : context->slabs)
			(void)hipFree(slab.base);
		for (host_database& db : context->databases)
		{
			if (!db.in_use)
				continue;
			(void)hipFree(db.d_runtime_headers);
			(void)hipFree(db.d_bound_clips);
			for (int tier = 0; tier < 2; ++tier)
			{
				(void)hipFree(db.d_bulk_data[tier]);
				(void)hipFree(db.d_patches[tier]);
				if (db.pinned_bulk_data[tier] != nullptr)
					(void)hipHostFree(db.pinned_bulk_data[tier]);
				if (db.pinned_patches[tier] != nullptr)
					(void)hipHostFree(db.pinned_patches[tier]);
			}
		}
		for (const aclhip_context::peer_mapping& mapping : context->peer_mappings)
			(void)hipIpcCloseMemHandle(mapping.base);		// buffers of other processes the caller never closed (aclhip_peer_close_buffer)
		if (context->d_clips != nullptr && context->table_is_virtual)
		{
			if (context->table_mapped_bytes != 0)
				(void)hipMemUnmap(context->d_clips, context->table_mapped_bytes);
			for (hipMemGenericAllocationHandle_t handle : context->table_handles)
				(void)hipMemRelease(handle);
			(void)hipMemAddressFree(context->d_clips, context->table_reserved_bytes);
		}
		else if (context->d_clips != nullptr)
			(void)hipFree(context->d_clips);
		if (context->d_rejected != nullptr)
			(void)hipFree(context->d_rejected);
		for (const aclhip_context::order_scratch& scratch : context->order_scratches)
		{
			(void)hipFree(scratch.bins);
			(void)hipFree(scratch.barrier);
			if (scratch.host_failed != nullptr)
				(void)hipHostFree(scratch.host_failed);
		}
	}
	delete context;
}
