// kernels_track.inl -- part of aclhip.hip (one translation unit; included there, in this order, not compiled on its own).
// decompress_track_kernel: single (instance, bone) requests.

	__global__ __launch_bounds__(k_block_size) void decompress_track_kernel(const device_clip* __restrict__ clips, uint32_t num_clips,
		const uint32_t* __restrict__ clip_ids, const float* __restrict__ sample_times, const uint32_t* __restrict__ track_indices, uint32_t num_instances,
		decode_params params, float4* __restrict__ transforms, unsigned long long* __restrict__ rejected_count)
	{
		const uint32_t instance = blockIdx.x * k_block_size + threadIdx.x;
		if (instance >= num_instances)
			return;

		const uint32_t clip_id = clip_ids[instance];
		if (clip_id >= num_clips || !is_transform_clip(clips[clip_id].flags))
		{
			atomicAdd(rejected_count, 1ull);
			return;
		}

		const device_clip& clip = clips[clip_id];
		const uint32_t track_index = track_indices[instance];
		if (track_index >= clip.num_tracks)
		{
			// invalid track index (decompression.transform.h:1766-1768); an empty clip lands here as well
			atomicAdd(rejected_count, 1ull);
			return;
		}

		const uint32_t rounding_policy = params.instance_rounding_policies != nullptr ? uint32_t(params.instance_rounding_policies[instance]) : uint32_t(params.rounding_policy);

		seek_state state;
		seek(clip, sample_times[instance], rounding_policy, params.looping_policy, state);

		// decompress_track_v0 folds a per track policy into the alpha and always interpolates (decompression.transform.h:1975-1983)
		float lerp_alpha = state.interpolation_alpha;
		if (params.per_track_rounding != 0)
		{
			uint32_t policy = rounding_policy;
			if (rounding_policy == k_round_per_track)
				policy = params.track_rounding_policies != nullptr ? params.track_rounding_policies[track_index] : k_round_none;
			lerp_alpha = apply_rounding_policy(lerp_alpha, policy);
		}

		// The reference sums the widths of every preceding animated sub-track to find this one's bits
		// (skip_*_groups + count_animated_group_bit_size, animated_track_cache.transform.h:1105-1192,1664-1707): O(track index).
		// The registration time plan already holds that prefix sum.
		const auto animated_lookup = [&](uint32_t ordinal)
		{
			const plan_entry plan0 = load_entry(clip.plan + size_t(state.segment_index[0]) * clip.num_animated, ordinal);
			const plan_entry plan1 = load_entry(clip.plan + size_t(state.segment_index[1]) * clip.num_animated, ordinal);
			const clip_range_entry clip_range = load_entry(clip.clip_ranges, ordinal);
			return decode_animated_sub_track<true, false>(state, plan0, plan1, clip_range, is_rotation_entry(clip_range),
				k_round_none, lerp_alpha, params.normalization, false);
		};

		for (uint32_t kind = 0; kind < 3; ++kind)
		{
			const uint32_t quad = track_index * 3u + kind;
			// base pose quad: constant (real W), animated (marker + ordinal) or default (marker)
			float4 value = load_quad(clip.base_pose, quad);
			const uint32_t marker = __float_as_uint(value.w);
			bool store = true;
			if (int32_t(marker) < 0 && (marker & k_quad_animated) != 0)
				value = animated_lookup(marker & k_quad_ordinal_mask);
			else if (int32_t(marker) < 0)
				value = resolve_quad(params, value, quad, store);
			else if (params.normalization == ACLHIP_NORMALIZE_ALWAYS && kind == 0)
				value = quat_normalize(value);		// constant_track_cache.transform.h:163-175
			if (store)
				store_streaming(&transforms[size_t(instance) * 3 + kind], f32x4{ value.x, value.y, value.z, value.w });
		}
	}
