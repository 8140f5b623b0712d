// host_consumers.inl -- part of aclhip.hip (one translation unit; included there, in this order, not compiled on its own).
// Host side: hierarchies and their walk schedules, pose consumer launches, host pointer convenience entry points.

// ---- pose consumers ------------------------------------------------------------------------------------------------

namespace
{
	// local_to_object_space (compression/transform_pose_utils.h:35-50) walks transforms in index order and needs parents first: any
	// order that keeps a parent ahead of its children gives the same bits. The consumer kernel takes up to P transforms per step,
	// P = 64 lanes / instances per workgroup, so the walk is scheduled on the host, once per hierarchy: at every step the P ready
	// transforms with the longest chain of descendants below them (Hu's algorithm: optimal for unit-time tasks on a forest). A
	// 100 bone character of 13 depths, 4-18 wide, takes 14 steps of 8 instead of 19 (12, its depth, at 16 per step).
	struct hierarchy_tree
	{
		std::vector<uint32_t> height;			// transforms on the longest chain from this one down to a leaf
		std::vector<uint32_t> first_child;		// [num_tracks + 1] into children
		std::vector<uint32_t> children;
		std::vector<uint8_t> is_root;
	};

	// false: transform out_misplaced does not follow its parent
	bool build_hierarchy_tree(const uint32_t* parent_indices, uint32_t num_tracks, hierarchy_tree& out, uint32_t& out_misplaced)
	{
		out.height.assign(num_tracks, 1);
		out.first_child.assign(size_t(num_tracks) + 1, 0);
		out.children.assign(num_tracks, 0);
		out.is_root.assign(num_tracks, 0);
		for (uint32_t i = 0; i < num_tracks; ++i)
		{
			// transform 0 is a root whatever its parent index says: the reference never reads it
			out.is_root[i] = (i == 0 || parent_indices[i] == ACLHIP_NO_PARENT) ? 1 : 0;
			if (out.is_root[i])
				continue;
			if (parent_indices[i] >= i)
			{
				out_misplaced = i;
				return false;
			}
			out.first_child[parent_indices[i] + 1]++;
		}
		for (uint32_t i = num_tracks; i-- > 1;)
			if (!out.is_root[i])
				out.height[parent_indices[i]] = std::max(out.height[parent_indices[i]], out.height[i] + 1);
		for (uint32_t i = 0; i < num_tracks; ++i)
			out.first_child[i + 1] += out.first_child[i];
		std::vector<uint32_t> cursor(out.first_child.begin(), out.first_child.end() - 1);
		for (uint32_t i = 1; i < num_tracks; ++i)
			if (!out.is_root[i])
				out.children[cursor[parent_indices[i]]++] = i;
		return true;
	}

	// out_transforms: every transform that has a parent, in the order it is computed; step s covers [out_step_end[s - 1], out_step_end[s])
	void schedule_hierarchy_walk(const hierarchy_tree& tree, uint32_t num_tracks, uint32_t transforms_per_step, std::vector<uint32_t>& out_step_end, std::vector<uint32_t>& out_transforms)
	{
		out_step_end.clear();
		out_transforms.clear();
		// ready transforms, the one with the longest chain below it (then the lowest index) on top
		const auto less_urgent = [&](uint32_t a, uint32_t b) { return tree.height[a] != tree.height[b] ? tree.height[a] < tree.height[b] : a > b; };
		std::vector<uint32_t> ready;
		for (uint32_t i = 0; i < num_tracks; ++i)
			if (tree.is_root[i])
				for (uint32_t c = tree.first_child[i]; c < tree.first_child[i + 1]; ++c)
					ready.push_back(tree.children[c]);
		std::make_heap(ready.begin(), ready.end(), less_urgent);
		std::vector<uint32_t> taken;
		while (!ready.empty())
		{
			taken.clear();
			while (!ready.empty() && taken.size() < transforms_per_step)
			{
				std::pop_heap(ready.begin(), ready.end(), less_urgent);
				taken.push_back(ready.back());
				ready.pop_back();
			}
			// their children become ready for the NEXT step
			for (uint32_t transform : taken)
			{
				out_transforms.push_back(transform);
				for (uint32_t c = tree.first_child[transform]; c < tree.first_child[transform + 1]; ++c)
				{
					ready.push_back(tree.children[c]);
					std::push_heap(ready.begin(), ready.end(), less_urgent);
				}
			}
			out_step_end.push_back(uint32_t(out_transforms.size()));
		}
	}
}

extern "C" aclhip_status aclhip_plan_hierarchy_walk(const uint32_t* parent_indices, uint32_t num_tracks, uint32_t transforms_per_step, uint32_t* out_steps, uint32_t* out_num_steps)
{
	if ((parent_indices == nullptr && num_tracks != 0) || out_num_steps == nullptr || transforms_per_step == 0)
		return ACLHIP_ERROR_INVALID_ARGUMENT;
	return guarded(nullptr, [&]() -> aclhip_status
	{
		hierarchy_tree tree;
		uint32_t misplaced = 0;
		if (!build_hierarchy_tree(parent_indices, num_tracks, tree, misplaced))
			return ACLHIP_ERROR_INVALID_ARGUMENT;
		std::vector<uint32_t> step_end, transforms;
		schedule_hierarchy_walk(tree, num_tracks, transforms_per_step, step_end, transforms);
		*out_num_steps = uint32_t(step_end.size());
		if (out_steps != nullptr)
		{
			std::fill(out_steps, out_steps + num_tracks, 0u);
			uint32_t begin = 0;
			for (uint32_t step = 0; step < step_end.size(); ++step)
			{
				for (uint32_t k = begin; k < step_end[step]; ++k)
					out_steps[transforms[k]] = step + 1;
				begin = step_end[step];
			}
		}
		return ACLHIP_OK;
	});
}

extern "C" aclhip_status aclhip_set_clip_hierarchy(aclhip_context* context, aclhip_clip clip, const uint32_t* parent_indices, uint32_t num_tracks);

// aclhip_set_clip_hierarchy with the parent indices the blob carried (read at registration: parse_clip_metadata)
extern "C" aclhip_status aclhip_set_clip_hierarchy_from_metadata(aclhip_context* context, aclhip_clip clip)
{
	if (context == nullptr)
		return ACLHIP_ERROR_INVALID_ARGUMENT;
	std::vector<uint32_t> parents;
	{
		std::shared_lock<std::shared_mutex> lock(context->mutex);
		if (clip >= context->clips.size() || !context->clips[clip].in_use)
			return fail(context, ACLHIP_ERROR_UNKNOWN_CLIP, "unknown clip handle %u", clip);
		if (context->clips[clip].metadata.has_parent_track_indices == 0)
			return fail(context, ACLHIP_ERROR_NO_METADATA, "clip %u was compressed without include_parent_track_indices: pass the hierarchy to aclhip_set_clip_hierarchy", clip);
		parents = context->clips[clip].metadata_parents;
	}
	return aclhip_set_clip_hierarchy(context, clip, parents.data(), uint32_t(parents.size()));
}

extern "C" aclhip_status aclhip_set_clip_hierarchy(aclhip_context* context, aclhip_clip clip, const uint32_t* parent_indices, uint32_t num_tracks)
{
	if (context == nullptr)
		return ACLHIP_ERROR_INVALID_ARGUMENT;
	if (parent_indices == nullptr && num_tracks != 0)
		return fail(context, ACLHIP_ERROR_INVALID_ARGUMENT, "null parent index list");

	return guarded(context, [&]() -> aclhip_status
	{
		std::lock_guard<std::shared_mutex> lock(context->mutex);
		if (clip >= context->clips.size() || !context->clips[clip].in_use)
			return fail(context, ACLHIP_ERROR_UNKNOWN_CLIP, "unknown clip handle %u", clip);
		host_clip& entry = context->clips[clip];
		if (entry.info.track_type != k_track_type_qvvf)
			return fail(context, ACLHIP_ERROR_INVALID_ARGUMENT, "clip %u is a scalar track list: no hierarchy", clip);
		if (entry.info.num_tracks != num_tracks)
			return fail(context, ACLHIP_ERROR_INVALID_ARGUMENT, "%u parent indices for a clip of %u tracks", num_tracks, entry.info.num_tracks);
		if (num_tracks > 0xFFFFu)
			return fail(context, ACLHIP_ERROR_INVALID_ARGUMENT, "%u transforms: the pose consumers end at about 3400", num_tracks);

		hierarchy_tree tree;
		uint32_t misplaced = 0;
		if (!build_hierarchy_tree(parent_indices, num_tracks, tree, misplaced))
			return fail(context, ACLHIP_ERROR_INVALID_ARGUMENT, "transform %u has parent %u: transforms must be sorted parent first", misplaced, parent_indices[misplaced]);
		const auto is_root = [&](uint32_t i) { return i == 0 || parent_indices[i] == ACLHIP_NO_PARENT; };

		// [{offset of the schedule, its steps, its words, 0} for 1, 2, 4, 8 instances per workgroup: one 16 byte scalar load tells a wave
		// all it needs to request the copy] then per schedule, 16 byte aligned and padded to whole 16 byte pieces (it travels to LDS by DMA):
		// num_steps | words of this schedule | step_end[num_steps] | transform | parent << 16, in step order (16 bits each: the
		// consumers' LDS images end at about 3400 transforms; every word of the copy a wave keeps in LDS costs residency)
		std::vector<uint32_t> image(16, 0);
		uint32_t max_schedule_words = 0;
		for (uint32_t log2_instances = 0; log2_instances < 4; ++log2_instances)
		{
			std::vector<uint32_t> step_end, pairs;
			schedule_hierarchy_walk(tree, num_tracks, 64u >> log2_instances, step_end, pairs);
			for (uint32_t& pair : pairs)
				pair |= parent_indices[pair] << 16;

			const uint32_t num_steps = uint32_t(step_end.size());
			const uint32_t header_words = 2 + num_steps;
			const uint32_t schedule_words = align_to_u32(header_words + uint32_t(pairs.size()), 4);
			const uint32_t offset = uint32_t(image.size());
			image[log2_instances * 4 + 0] = offset;
			image[log2_instances * 4 + 1] = num_steps;
			image[log2_instances * 4 + 2] = schedule_words;
			image.resize(size_t(offset) + schedule_words, 0);
			image[offset + 0] = num_steps;
			image[offset + 1] = schedule_words;
			std::copy(step_end.begin(), step_end.end(), image.begin() + offset + 2);
			std::copy(pairs.begin(), pairs.end(), image.begin() + offset + header_words);
			max_schedule_words = std::max(max_schedule_words, schedule_words);
		}

		device_guard guard(context->device);

		// an identical hierarchy (another clip of the same skeleton) is already on the device?
		const std::vector<uint32_t> canonical = [&]()
		{
			std::vector<uint32_t> parents(parent_indices, parent_indices + num_tracks);
			for (uint32_t i = 0; i < num_tracks; ++i)
				if (is_root(i))
					parents[i] = ACLHIP_NO_PARENT;
			return parents;
		}();
		// (first: what this recycles may be the very image looked for below -- its last user retired it a moment ago -- and recycling
		// erases entries of the list the search walks. Round 3 searched first and could pick up a dangling entry: found by
		// tests/test_gpu_lifetime.py once the timing of its launches changed)
		collect_retired(context, false);

		aclhip_context::hierarchy_image* shared = nullptr;
		for (aclhip_context::hierarchy_image& candidate : context->hierarchies)
			if (candidate.parents == canonical)
				shared = &candidate;
		uint32_t* d_hierarchy = shared != nullptr ? shared->d_image : nullptr;
		size_t staging_used = 0;
		bool uploaded = true;
		if (shared == nullptr)
		{
			// (a piece of a clip slab, uploaded on the context's copy stream: no allocation call and no copy that would stall the device)
			d_hierarchy = reinterpret_cast<uint32_t*>(allocate_clip_memory(context, image.size() * sizeof(uint32_t)));
			if (d_hierarchy == nullptr)
				return fail(context, ACLHIP_ERROR_OUT_OF_MEMORY, "allocating %zu bytes for the hierarchy failed", image.size() * sizeof(uint32_t));
			uploaded = stage_upload(context, d_hierarchy, image.data(), image.size() * sizeof(uint32_t), staging_used);
		}
		// launches in flight may still walk the hierarchy that is being replaced: it is retired, not freed (see retire())
		uploaded = uploaded && stage_upload(context, reinterpret_cast<uint8_t*>(context->d_clips + clip) + offsetof(device_clip, hierarchy), &d_hierarchy, sizeof(d_hierarchy), staging_used)
			&& finish_uploads(context);
		if (!uploaded)
		{
			if (shared == nullptr)
				free_clip_memory(context, d_hierarchy);
			return fail(context, ACLHIP_ERROR_DEVICE, "uploading the hierarchy failed");
		}
		if (shared != nullptr)
			shared->num_users++;
		else
		{
			aclhip_context::hierarchy_image created;
			created.parents = canonical;
			created.d_image = d_hierarchy;
			created.num_users = 1;
			context->hierarchies.push_back(std::move(created));
		}
		if (entry.d_hierarchy != nullptr)
		{
			aclhip_context::retired_item item;
			item.hierarchy = entry.d_hierarchy;
			retire(context, std::move(item));
		}
		entry.d_hierarchy = d_hierarchy;
		const bool held_maximum = entry.hierarchy_words != 0 && entry.hierarchy_words == context->max_hierarchy_words;
		entry.hierarchy_words = max_schedule_words;
		context->max_hierarchy_words = std::max(context->max_hierarchy_words, max_schedule_words);
		if (held_maximum && max_schedule_words < context->max_hierarchy_words)
			recompute_launch_maxima(context);
		return ACLHIP_OK;
	});
}

namespace
{
	aclhip_status launch_consumers(aclhip_context* context, const aclhip_clip* clips, const float* sample_times, uint32_t num_instances,
		const decode_params& params, const aclhip_pose_consumers& consumers, void* poses, uint64_t pose_stride_bytes, hipStream_t stream)
	{
		if (consumers.additive_format > ACLHIP_ADDITIVE_ADDITIVE1)
			return fail(context, ACLHIP_ERROR_INVALID_ARGUMENT, "unknown additive format %u", consumers.additive_format);
		const bool has_base = consumers.additive_format != ACLHIP_ADDITIVE_NONE;
		const bool base_is_clip = has_base && consumers.base_clips != nullptr;
		if (base_is_clip && consumers.base_sample_times == nullptr)
			return fail(context, ACLHIP_ERROR_INVALID_ARGUMENT, "base clips without base sample times");
		if (has_base && !base_is_clip && (consumers.base_poses == nullptr || (consumers.base_pose_stride_bytes & 15u) != 0 || (reinterpret_cast<uintptr_t>(consumers.base_poses) & 15u) != 0))
			return fail(context, ACLHIP_ERROR_INVALID_ARGUMENT, "an additive format needs base clips or a 16 byte aligned base pose buffer");
		const bool blend = consumers.num_blend_clips > 1;
		if ((consumers.flags & ~ACLHIP_CONSUMERS_FAST) != 0)
			return fail(context, ACLHIP_ERROR_INVALID_ARGUMENT, "unknown pose consumer flags 0x%x", consumers.flags);
		if (consumers.num_blend_clips > ACLHIP_MAX_BLEND_CLIPS)
			return fail(context, ACLHIP_ERROR_INVALID_ARGUMENT, "a blend of %u clips: at most %u", consumers.num_blend_clips, ACLHIP_MAX_BLEND_CLIPS);
		if (blend && (consumers.blend_clips == nullptr || consumers.blend_sample_times == nullptr || consumers.blend_weights == nullptr))
			return fail(context, ACLHIP_ERROR_INVALID_ARGUMENT, "a blend needs blend_clips, blend_sample_times and blend_weights");
		// a consumer needs every sub-track of the pose: the track_writer's own defaults (what the resolved pose image holds)
		if (params.standard_defaults == 0 || params.per_track_rounding != 0)
			return fail(context, ACLHIP_ERROR_INVALID_ARGUMENT, "pose consumers take the track_writer's default sub-track modes, no per track rounding, normalization != always");

		std::shared_lock<std::shared_mutex> lock(context->mutex);		// see launch_tracks
		note_launch_stream(context, stream);

		// one wave per instance, the whole pose (its base, its hierarchy) in LDS; as many instances per workgroup (a power of two, at
		// most 8, 4 unless told otherwise: measured best) as leave room for three workgroups per CU: the object space walk packs its lanes with instances of one workgroup
		// No clip with a scale other than 1 registered, no base to combine with: every scale of every pose is 1, in local and in object
		// space -- the LDS images hold rotation | translation (32 of a transform's 48 bytes: half as many poses again per CU) and the
		// scales are written on the way out
		const bool unit_scale = !has_base && !blend && consumers.object_space != 0 && context->num_scaled_clips == 0 && path_knob("ACLHIP_CONSUMER_KEEP_SCALE") == nullptr;
		// (sized for the BATCH like every pose launch: no pose of it is larger than its row, pose_launch_shape_of in host_launch.inl --
		// one 3 500-bone asset in the registry does not take object space away from the 100-bone characters)
		const uint32_t batch_quads = batch_pose_quads(context, ACLHIP_LAYOUT_QVV48, pose_stride_bytes);
		const uint32_t image_quads = unit_scale ? batch_quads / 3 * 2 : batch_quads;
		// exactly the transforms a pose row holds (or the largest registered clip has), not a quad more: the kernel's "does the pose fit
		// its image" test is also its "does the pose fit its row" test (every quad is addressed on its own: no granularity needed)
		const uint32_t lds_quads_per_image = std::max<uint32_t>(image_quads, 1);
		// additive0 / additive1 combine sub-track with sub-track: the base clip is decoded into the instance's image and the additive clip
		// onto it by one wave; the relative format (a qvv_mul) needs both poses whole: a second wave, a second image
		// (a blend accumulates its clips in the instance's image before anything else happens to it: its base clip gets a wave and an image of its own)
		const bool fused_base = base_is_clip && !blend && consumers.additive_format != ACLHIP_ADDITIVE_RELATIVE && lab_knob("ACLHIP_CONSUMER_TWO_IMAGES") == nullptr;
		const bool two_waves = base_is_clip && !fused_base;
		// (measurement knob: ACLHIP_CONSUMER_LDS_PAD bytes between the instances' images -- the walk's lanes touch the same quad of all of a
		// workgroup's images at once, and images a multiple of 128 bytes apart put those on the same LDS banks)
		// 16 bytes: the four images of a workgroup then start on different banks (round 4: 88.8 -> 86.9 us, 90.6 -> 83.7 us with ACLHIP_CONSUMERS_FAST;
		// 32 the same, 64 less, 0 what rounds 2 and 3 measured)
		static const size_t lds_pad = []() { const char* value = lab_knob("ACLHIP_CONSUMER_LDS_PAD"); return value != nullptr ? size_t(std::atol(value)) & ~size_t(15) : size_t(16); }();
		const size_t lds_bytes_per_instance = size_t(lds_quads_per_image) * 16 * (two_waves ? 2 : 1) + lds_pad;
		// a walk schedule of T transforms: 2 words + a step end per step + a pair per transform with a parent, at most 2 + 2 T words
		const uint32_t lds_schedule_words = consumers.object_space != 0 ? align_to_u32(std::max<uint32_t>(std::min<uint32_t>(context->max_hierarchy_words, 2 + 2 * (batch_quads / 3) + 3), 4), 4) : 0;
		const size_t lds_schedule_bytes = size_t(lds_schedule_words) * sizeof(uint32_t);
		// what a workgroup may ask for on top of the kernel's static words (consumer_walk_slots, kernels_consumers.inl)
		constexpr size_t k_lds_bytes = 160 * 1024 - ((sizeof(consumer_walk_slots) + 127) / 128) * 128;
		if (lds_bytes_per_instance + lds_schedule_bytes > k_lds_bytes)
			return fail(context, ACLHIP_ERROR_INVALID_ARGUMENT, "poses of %u transforms (the pose stride, the largest registered clip): too large for the pose consumers (%zu bytes of LDS per instance)", batch_quads / 3, lds_bytes_per_instance + lds_schedule_bytes);
		uint32_t log2_instances_per_block = 2;
		if (const char* forced = lab_knob("ACLHIP_CONSUMER_LOG2_INSTANCES"))
			log2_instances_per_block = std::min<uint32_t>(uint32_t(forced[0] - '0'), 3);
		while (log2_instances_per_block != 0 && (lds_bytes_per_instance << log2_instances_per_block) + lds_schedule_bytes > k_lds_bytes / 3)
			log2_instances_per_block--;
		const uint32_t instances_per_block = 1u << log2_instances_per_block;
		const uint32_t waves_per_block = instances_per_block * (two_waves ? 2 : 1);
		const uint32_t num_blocks = (num_instances + instances_per_block - 1) / instances_per_block;
		const size_t lds_bytes = lds_bytes_per_instance * instances_per_block + lds_schedule_bytes;

		consumer_params device_consumers;
		device_consumers.base_clip_ids = base_is_clip ? consumers.base_clips : nullptr;
		device_consumers.base_sample_times = base_is_clip ? consumers.base_sample_times : nullptr;
		device_consumers.base_poses = has_base && !base_is_clip ? static_cast<const uint8_t*>(consumers.base_poses) : nullptr;
		device_consumers.base_pose_stride_bytes = consumers.base_pose_stride_bytes;
		device_consumers.additive_format = consumers.additive_format;
		device_consumers.object_space = consumers.object_space != 0 ? 1 : 0;
		device_consumers.blend_clip_ids = blend ? consumers.blend_clips : nullptr;
		device_consumers.blend_sample_times = blend ? consumers.blend_sample_times : nullptr;
		device_consumers.blend_weights = blend ? consumers.blend_weights : nullptr;
		device_consumers.num_blend_clips = blend ? consumers.num_blend_clips : 0;

		// one instantiation per (object space, kind of base, rotation | translation images)
		const uint32_t base_kind = !has_base ? k_consumer_base_none : (!base_is_clip ? k_consumer_base_buffer : (fused_base ? k_consumer_base_fused : k_consumer_base_second_wave));
		typedef void (*consumer_kernel)(const device_clip*, uint32_t, const uint32_t*, const float*, uint32_t, decode_params, consumer_params, uint8_t*, uint64_t, uint32_t, uint32_t, uint32_t, unsigned long long*);
		// rtm::qvv_mul's matrix route (negative scales) is compiled into the launches that can meet one: a registered clip whose scale
		// sub-tracks may decode below zero, or a base the library knows nothing about (a caller's pose buffer)
		const bool mirrored = context->num_negative_scale_clips != 0 || (has_base && !base_is_clip);
		static const consumer_kernel kernels[2][2][4] =
		{
			{
				{ decompress_poses_consumer_kernel<false, k_consumer_base_none, false, false>, decompress_poses_consumer_kernel<false, k_consumer_base_buffer, false, false>,
				  decompress_poses_consumer_kernel<false, k_consumer_base_second_wave, false, false>, decompress_poses_consumer_kernel<false, k_consumer_base_fused, false, false> },
				{ decompress_poses_consumer_kernel<true, k_consumer_base_none, false, false>, decompress_poses_consumer_kernel<true, k_consumer_base_buffer, false, false>,
				  decompress_poses_consumer_kernel<true, k_consumer_base_second_wave, false, false>, decompress_poses_consumer_kernel<true, k_consumer_base_fused, false, false> },
			},
			{
				// (no object space: only the relative format multiplies transforms -- a base buffer or a second wave's image)
				{ decompress_poses_consumer_kernel<false, k_consumer_base_none, false, false>, decompress_poses_consumer_kernel<false, k_consumer_base_buffer, false, true>,
				  decompress_poses_consumer_kernel<false, k_consumer_base_second_wave, false, true>, decompress_poses_consumer_kernel<false, k_consumer_base_fused, false, false> },
				{ decompress_poses_consumer_kernel<true, k_consumer_base_none, false, true>, decompress_poses_consumer_kernel<true, k_consumer_base_buffer, false, true>,
				  decompress_poses_consumer_kernel<true, k_consumer_base_second_wave, false, true>, decompress_poses_consumer_kernel<true, k_consumer_base_fused, false, true> },
			},
		};
		// the same with a blend in front (never fused, never rotation | translation images)
		static const consumer_kernel blend_kernels[2][2][3] =
		{
			{
				{ decompress_poses_consumer_kernel<false, k_consumer_base_none, false, false, true>, decompress_poses_consumer_kernel<false, k_consumer_base_buffer, false, true, true>,
				  decompress_poses_consumer_kernel<false, k_consumer_base_second_wave, false, false, true> },
				{ decompress_poses_consumer_kernel<true, k_consumer_base_none, false, false, true>, decompress_poses_consumer_kernel<true, k_consumer_base_buffer, false, true, true>,
				  decompress_poses_consumer_kernel<true, k_consumer_base_second_wave, false, false, true> },
			},
			{
				{ decompress_poses_consumer_kernel<false, k_consumer_base_none, false, false, true>, decompress_poses_consumer_kernel<false, k_consumer_base_buffer, false, true, true>,
				  decompress_poses_consumer_kernel<false, k_consumer_base_second_wave, false, true, true> },
				{ decompress_poses_consumer_kernel<true, k_consumer_base_none, false, true, true>, decompress_poses_consumer_kernel<true, k_consumer_base_buffer, false, true, true>,
				  decompress_poses_consumer_kernel<true, k_consumer_base_second_wave, false, true, true> },
			},
		};
		// ACLHIP_CONSUMERS_FAST: object space launches without a blend, in the hardware's 1 ulp arithmetic (include/aclhip.h)
		const bool fast = (consumers.flags & ACLHIP_CONSUMERS_FAST) != 0 && consumers.object_space != 0 && !blend;
		static const consumer_kernel fast_kernels[2][4] =
		{
			{ decompress_poses_consumer_kernel<true, k_consumer_base_none, false, false, false, true>, decompress_poses_consumer_kernel<true, k_consumer_base_buffer, false, true, false, true>,
			  decompress_poses_consumer_kernel<true, k_consumer_base_second_wave, false, false, false, true>, decompress_poses_consumer_kernel<true, k_consumer_base_fused, false, false, false, true> },
			{ decompress_poses_consumer_kernel<true, k_consumer_base_none, false, true, false, true>, decompress_poses_consumer_kernel<true, k_consumer_base_buffer, false, true, false, true>,
			  decompress_poses_consumer_kernel<true, k_consumer_base_second_wave, false, true, false, true>, decompress_poses_consumer_kernel<true, k_consumer_base_fused, false, true, false, true> },
		};
		const consumer_kernel kernel = fast ? (unit_scale ? decompress_poses_consumer_kernel<true, k_consumer_base_none, true, false, false, true> : fast_kernels[mirrored ? 1 : 0][base_kind])
			: unit_scale ? decompress_poses_consumer_kernel<true, k_consumer_base_none, true, false>
			: (blend ? blend_kernels[mirrored ? 1 : 0][consumers.object_space != 0 ? 1 : 0][base_kind] : kernels[mirrored ? 1 : 0][consumers.object_space != 0 ? 1 : 0][base_kind]);
		if (lds_bytes > 64 * 1024 - 128)		// above the default limit
			ACLHIP_CHECK_HIP(context, hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, int(k_lds_bytes)));
		hipLaunchKernelGGL(kernel, dim3(num_blocks), dim3(waves_per_block * k_wave_size), lds_bytes, stream,
			context->d_clips, context->d_clips_capacity, clips, sample_times, num_instances, params, device_consumers,
			static_cast<uint8_t*>(poses), pose_stride_bytes, lds_quads_per_image, uint32_t(lds_bytes_per_instance), log2_instances_per_block | (lds_schedule_words << 8), context->d_rejected);
		ACLHIP_CHECK_HIP(context, hipGetLastError());
		return ACLHIP_OK;
	}
}

extern "C" aclhip_status aclhip_decompress_poses_batch(aclhip_context* context, const aclhip_clip* clips, const float* sample_times, uint32_t num_instances,
	const aclhip_decompress_params* params, const aclhip_pose_consumers* consumers, void* poses, uint64_t pose_stride_bytes, void* stream)
{
	aclhip_status status = check_batch_arguments(context, clips, sample_times, num_instances, poses, pose_stride_bytes);
	if (status != ACLHIP_OK)
		return status;
	if (consumers == nullptr)
		return fail(context, ACLHIP_ERROR_INVALID_ARGUMENT, "null consumers");
	if (num_instances == 0)
		return ACLHIP_OK;

	decode_params device_params;
	status = resolve_params(context, params, device_params);
	if (status != ACLHIP_OK)
		return status;

	device_guard guard(context->device);
	return launch_consumers(context, clips, sample_times, num_instances, device_params, *consumers, poses, pose_stride_bytes, static_cast<hipStream_t>(stream));
}

namespace
{
	// Host pointer convenience path: upload, launch, download, synchronously
	aclhip_status decompress_host(aclhip_context* context, const aclhip_clip* clips, const float* sample_times, const uint32_t* track_indices, uint32_t num_instances,
		const aclhip_decompress_params* params, uint32_t default_values_count, void* out, uint64_t out_stride_bytes, uint64_t out_row_bytes,
		const aclhip_pose_consumers* consumers = nullptr, const aclhip_output_desc* output = nullptr)
	{
		if (context == nullptr)
			return ACLHIP_ERROR_INVALID_ARGUMENT;
		if (num_instances == 0)
			return ACLHIP_OK;
		if (clips == nullptr || sample_times == nullptr || out == nullptr)
			return fail(context, ACLHIP_ERROR_INVALID_ARGUMENT, "null instance list or output buffer");
		const uint32_t bytes_per_track = output != nullptr ? aclhip_layout_bytes_per_track(output->layout) : 48u;
		if (bytes_per_track == 0)
			return fail(context, ACLHIP_ERROR_INVALID_ARGUMENT, "unknown pose layout %u", output->layout);

		aclhip_decompress_params local;
		if (params != nullptr) local = *params; else aclhip_default_params(&local);

		device_guard guard(context->device);

		uint32_t max_tracks = 0;
		{
			std::lock_guard<std::shared_mutex> lock(context->mutex);
			for (uint32_t i = 0; i < num_instances; ++i)
				if (clips[i] < context->clips.size() && context->clips[clips[i]].in_use)
					max_tracks = std::max(max_tracks, context->clips[clips[i]].info.num_tracks);
		}

		const bool single_track = track_indices != nullptr;
		const uint64_t device_stride = single_track ? 48 : std::max<uint64_t>((uint64_t(max_tracks) * bytes_per_track + 15) & ~uint64_t(15), 16);
		if (!single_track && out_row_bytes == 0)
			out_row_bytes = uint64_t(max_tracks) * bytes_per_track;

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

		void* d_clip_ids = nullptr; void* d_times = nullptr; void* d_tracks = nullptr; void* d_out = nullptr;
		void* d_defaults = nullptr; void* d_track_policies = nullptr; void* d_instance_policies = nullptr;
		bool ok = upload(clips, sizeof(uint32_t) * num_instances, &d_clip_ids) && upload(sample_times, sizeof(float) * num_instances, &d_times);
		if (ok && single_track)
			ok = upload(track_indices, sizeof(uint32_t) * num_instances, &d_tracks);
		ok = ok && upload(nullptr, device_stride * num_instances, &d_out);
		aclhip_output_desc local_output = {};
		if (output != nullptr)
		{
			// (the same refusals apply_output_desc makes for the device entry points, BEFORE anything is uploaded: a descriptor with one of
			// the two mask arrays used to be decoded here without masks)
			if ((output->instance_masks != nullptr) != (output->mask_table != nullptr))
			{
				release();
				return fail(context, ACLHIP_ERROR_INVALID_ARGUMENT, "instance_masks and mask_table come together");
			}
			if (output->instance_masks != nullptr && output->mask_stride == 0)
			{
				release();
				return fail(context, ACLHIP_ERROR_INVALID_ARGUMENT, "mask_stride: the bytes of one mask of mask_table (at least the tracks of the largest clip of the batch)");
			}
			local_output = *output;
			void* d_rows = nullptr;
			if (ok && output->rows != nullptr)
				ok = upload(output->rows, sizeof(uint32_t) * num_instances, &d_rows);
			local_output.rows = static_cast<const uint32_t*>(d_rows);
			void* d_skip_tracks = nullptr;
			if (ok && output->skip_tracks != nullptr)
				ok = upload(output->skip_tracks, std::max<uint32_t>(max_tracks, 1), &d_skip_tracks);
			local_output.skip_tracks = static_cast<const uint8_t*>(d_skip_tracks);
			// per instance writer decisions: the masks the instances name (as many as the largest index says), their track counts
			void* d_instance_masks = nullptr; void* d_mask_table = nullptr; void* d_track_counts = nullptr;
			if (ok && output->instance_masks != nullptr && output->mask_table != nullptr)
			{
				uint32_t num_masks = 0;
				for (uint32_t i = 0; i < num_instances; ++i)
					num_masks = std::max<uint32_t>(num_masks, uint32_t(output->instance_masks[i]) + 1);
				ok = upload(output->instance_masks, num_instances, &d_instance_masks) && upload(output->mask_table, size_t(num_masks) * output->mask_stride, &d_mask_table);
			}
			if (ok && output->instance_track_counts != nullptr)
				ok = upload(output->instance_track_counts, sizeof(uint32_t) * num_instances, &d_track_counts);
			local_output.instance_masks = output->instance_masks != nullptr ? static_cast<const uint8_t*>(d_instance_masks) : nullptr;
			local_output.mask_table = output->mask_table != nullptr ? static_cast<const uint8_t*>(d_mask_table) : nullptr;
			local_output.instance_track_counts = static_cast<const uint32_t*>(d_track_counts);
		}
		if (ok && local.default_values != nullptr)
			ok = upload(local.default_values, size_t(std::max<uint32_t>(default_values_count, 1)) * 48, &d_defaults);
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

		aclhip_pose_consumers local_consumers = {};
		if (consumers != nullptr)
		{
			local_consumers = *consumers;
			void* d_base_clips = nullptr; void* d_base_times = nullptr; void* d_base_poses = nullptr;
			if (ok && consumers->base_clips != nullptr)
				ok = upload(consumers->base_clips, sizeof(uint32_t) * num_instances, &d_base_clips);
			if (ok && consumers->base_sample_times != nullptr)
				ok = upload(consumers->base_sample_times, sizeof(float) * num_instances, &d_base_times);
			if (ok && consumers->base_poses != nullptr && consumers->base_clips == nullptr && consumers->additive_format != ACLHIP_ADDITIVE_NONE)
			{
				if (consumers->base_pose_stride_bytes < uint64_t(max_tracks) * 48)
				{
					release();
					return fail(context, ACLHIP_ERROR_INVALID_ARGUMENT, "base pose stride %llu is smaller than a pose of %u transforms", (unsigned long long)consumers->base_pose_stride_bytes, max_tracks);
				}
				ok = upload(consumers->base_poses, size_t(consumers->base_pose_stride_bytes) * num_instances, &d_base_poses);
			}
			local_consumers.base_clips = static_cast<const aclhip_clip*>(d_base_clips);
			local_consumers.base_sample_times = static_cast<const float*>(d_base_times);
			local_consumers.base_poses = d_base_poses;
			if (consumers->num_blend_clips > 1 && consumers->num_blend_clips <= ACLHIP_MAX_BLEND_CLIPS
				&& consumers->blend_clips != nullptr && consumers->blend_sample_times != nullptr && consumers->blend_weights != nullptr)
			{
				const size_t others = size_t(num_instances) * (consumers->num_blend_clips - 1);
				void* d_blend_clips = nullptr; void* d_blend_times = nullptr; void* d_blend_weights = nullptr;
				ok = ok && upload(consumers->blend_clips, sizeof(uint32_t) * others, &d_blend_clips) && upload(consumers->blend_sample_times, sizeof(float) * others, &d_blend_times)
					&& upload(consumers->blend_weights, sizeof(float) * size_t(num_instances) * consumers->num_blend_clips, &d_blend_weights);
				local_consumers.blend_clips = static_cast<const aclhip_clip*>(d_blend_clips);
				local_consumers.blend_sample_times = static_cast<const float*>(d_blend_times);
				local_consumers.blend_weights = static_cast<const float*>(d_blend_weights);
			}
		}
		if (!ok)
		{
			release();
			return fail(context, ACLHIP_ERROR_DEVICE, "staging the batch on the device failed");
		}

		// Bytes the decode does not write (skipped defaults, tracks beyond a smaller clip's count, rejected instances) must keep
		// what the caller had there: round trip the caller's buffer
		{
			const hipError_t copy_status = hipMemcpy2D(d_out, device_stride, out, out_stride_bytes, std::min<uint64_t>(out_row_bytes, device_stride), num_instances, hipMemcpyHostToDevice);
			if (copy_status != hipSuccess)
			{
				release();
				return fail(context, ACLHIP_ERROR_DEVICE, "uploading the caller's pose buffer failed: %s", hipGetErrorString(copy_status));
			}
		}

		local.default_values = static_cast<const float*>(d_defaults);
		local.track_rounding_policies = static_cast<const uint8_t*>(d_track_policies);
		local.instance_rounding_policies = static_cast<const uint8_t*>(d_instance_policies);
		local.instance_looping_policies = static_cast<const uint8_t*>(d_instance_looping);
		if (local.track_rounding_table != nullptr && local.instance_rounding_tables != nullptr)
		{
			local.track_rounding_table = static_cast<const uint8_t*>(d_rounding_table);
			local.instance_rounding_tables = static_cast<const uint8_t*>(d_rounding_tables_of);
		}

		aclhip_status status;
		if (single_track)
			status = aclhip_decompress_track_batch(context, static_cast<const aclhip_clip*>(d_clip_ids), static_cast<const float*>(d_times), static_cast<const uint32_t*>(d_tracks), num_instances, &local, d_out, work_stream);
		else if (consumers != nullptr)
			status = aclhip_decompress_poses_batch(context, static_cast<const aclhip_clip*>(d_clip_ids), static_cast<const float*>(d_times), num_instances, &local, &local_consumers, d_out, device_stride, work_stream);
		else
			status = aclhip_decompress_tracks_batch_out(context, static_cast<const aclhip_clip*>(d_clip_ids), static_cast<const float*>(d_times), num_instances, &local,
				output != nullptr ? &local_output : nullptr, d_out, device_stride, work_stream);

		if (status == ACLHIP_OK)
		{
			hipError_t copy_status = hipStreamSynchronize(work_stream);
			if (copy_status == hipSuccess)
				copy_status = hipMemcpy2D(out, out_stride_bytes, d_out, device_stride, std::min<uint64_t>(out_row_bytes, device_stride), num_instances, hipMemcpyDeviceToHost);
			if (copy_status != hipSuccess)
				status = fail(context, ACLHIP_ERROR_DEVICE, "downloading the poses failed: %s", hipGetErrorString(copy_status));
		}

		release();
		return status;
	}
}

extern "C" aclhip_status aclhip_decompress_tracks_host(aclhip_context* context, const aclhip_clip* clips, const float* sample_times, uint32_t num_instances,
	const aclhip_decompress_params* params, uint32_t default_values_count, void* poses, uint64_t pose_stride_bytes)
{
	return decompress_host(context, clips, sample_times, nullptr, num_instances, params, default_values_count, poses, pose_stride_bytes, 0);
}

extern "C" aclhip_status aclhip_decompress_tracks_host_out(aclhip_context* context, const aclhip_clip* clips, const float* sample_times, uint32_t num_instances,
	const aclhip_decompress_params* params, uint32_t default_values_count, const aclhip_output_desc* output, void* poses, uint64_t pose_stride_bytes)
{
	return decompress_host(context, clips, sample_times, nullptr, num_instances, params, default_values_count, poses, pose_stride_bytes, 0, nullptr, output);
}

extern "C" aclhip_status aclhip_decompress_poses_host(aclhip_context* context, const aclhip_clip* clips, const float* sample_times, uint32_t num_instances,
	const aclhip_decompress_params* params, const aclhip_pose_consumers* consumers, void* poses, uint64_t pose_stride_bytes)
{
	if (consumers == nullptr)
		return context != nullptr ? fail(context, ACLHIP_ERROR_INVALID_ARGUMENT, "null consumers") : ACLHIP_ERROR_INVALID_ARGUMENT;
	if (params != nullptr && (params->default_values != nullptr || params->track_rounding_policies != nullptr || params->track_rounding_table != nullptr))
		return context != nullptr ? fail(context, ACLHIP_ERROR_INVALID_ARGUMENT, "pose consumers take the track_writer's default sub-track modes, no per track rounding") : ACLHIP_ERROR_INVALID_ARGUMENT;
	return decompress_host(context, clips, sample_times, nullptr, num_instances, params, 0, poses, pose_stride_bytes, 0, consumers);
}

extern "C" aclhip_status aclhip_decompress_track_host(aclhip_context* context, const aclhip_clip* clips, const float* sample_times, const uint32_t* track_indices,
	uint32_t num_instances, const aclhip_decompress_params* params, uint32_t default_values_count, void* transforms)
{
	if (track_indices == nullptr)
		return context != nullptr ? fail(context, ACLHIP_ERROR_INVALID_ARGUMENT, "null track index list") : ACLHIP_ERROR_INVALID_ARGUMENT;
	return decompress_host(context, clips, sample_times, track_indices, num_instances, params, default_values_count, transforms, 48, 48);
}

#if defined(ACLHIP_EXP_PHASE_TIMES)
// measurement aid, see kernels_consumers.inl
extern "C" int aclhip_debug_read_phase_times(unsigned long long* out, uint32_t count)
{
	return int(hipMemcpyFromSymbol(out, HIP_SYMBOL(aclhip::phase_times), size_t(count) * sizeof(unsigned long long), 0, hipMemcpyDeviceToHost));
}
#endif
