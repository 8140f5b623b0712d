// kernels_consumers.inl -- part of aclhip.hip (one translation unit; included there, in this order, not compiled on its own).
// decompress_poses_consumer_kernel: the decode followed by additive apply and local -> object space while the pose is in LDS.

	// ---- pose consumers -----------------------------------------------------------------------------------------------
	// Decodes the whole local pose of one clip instance into an LDS image (image[0] = quad 0), window by window like the pose kernels
	// but in ONE wave, because what follows needs every transform of the pose. Common-case settings only (see launch_consumers).
	template<bool kFastMath = false, class image_writer_type>
	__device__ __forceinline__ void decode_animated_into_image(const device_clip& clip, float sample_time, uint32_t rounding_policy, const decode_params& params,
		uint32_t lane, image_writer_type write_to_image)
	{
		seek_state state;
		seek(clip, sample_time, rounding_policy, params.looping_policy, state);

		const uint32_t num_quads = clip.num_tracks * 3u;
		if (num_quads <= k_image_chunk_quads)
			decode_window_sub_tracks_into<false, false, (kFastMath ? 1u : 0u)>(window_tables_of(clip), state, params, rounding_policy, params.normalization, 0, clip.num_animated, lane, write_to_image);
		else
		{
			const uint32_t num_windows = (num_quads + k_image_chunk_quads - 1) / k_image_chunk_quads;
			for (uint32_t window = 0; window < num_windows; ++window)
			{
				const uint32_t first_ordinal = as_constant(clip.image_chunks)[window];
				const uint32_t end_ordinal = as_constant(clip.image_chunks)[window + 1];
				decode_window_sub_tracks_into<false, false, (kFastMath ? 1u : 0u)>(window_tables_of(clip), state, params, rounding_policy, params.normalization, first_ordinal, end_ordinal, lane, write_to_image);
			}
		}
	}

	template<bool kFastMath = false>
	__device__ __forceinline__ void decode_pose_into_image(const device_clip& clip, float sample_time, uint32_t rounding_policy, const decode_params& params,
		uint32_t lane, f32x4* image)
	{
		const uint32_t num_quads = clip.num_tracks * 3u;
		{
			const ACLHIP_CONSTANT f32x4* source = (const ACLHIP_CONSTANT f32x4*)clip.resolved_pose;
			for (uint32_t base = 0; base < num_quads; base += k_wave_size)
			{
				if (base + lane < num_quads)
					__builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(source + base + lane),
						(__attribute__((address_space(3))) void*)(image + base), 16, 0, 0);
			}
		}
		decode_animated_into_image<kFastMath>(clip, sample_time, rounding_policy, params, lane, qvv48_image_writer{ image, 0, 0xFFFFFFFFu });
	}

	// The pose without its scales (every one of them is 1): rotation | translation, 32 bytes per transform, like the QV32 output layout
	template<bool kFastMath = false>
	__device__ __forceinline__ void decode_unit_scale_pose_into_image(const device_clip& clip, float sample_time, uint32_t rounding_policy, const decode_params& params,
		uint32_t lane, f32x4* image)
	{
		const uint32_t num_pieces = clip.num_tracks * 2u;
		const ACLHIP_CONSTANT f32x4* resolved = (const ACLHIP_CONSTANT f32x4*)clip.resolved_pose;
		for (uint32_t base = 0; base < num_pieces; base += k_wave_size)
		{
			const uint32_t piece = base + lane;
			if (piece < num_pieces)
				__builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(resolved + (piece >> 1) * 3u + (piece & 1u)),
					(__attribute__((address_space(3))) void*)(image + base), 16, 0, 0);
		}
		decode_animated_into_image<kFastMath>(clip, sample_time, rounding_policy, params, lane, compact_image_writer<ACLHIP_LAYOUT_QV32>{ reinterpret_cast<float*>(image), 0, 0xFFFFFFFFu });
	}

	// transform_add0 / transform_add1 (core/additive_utils.h:128-142) one sub-track at a time: unlike the relative format (a qvv_mul)
	// they combine rotation with rotation, translation with translation and scale with scale. kind: 0 rotation, 1 translation, 2 scale
	template<bool kFastMath = false>
	__device__ __forceinline__ f32x4 apply_additive_sub_track(uint32_t additive_format, uint32_t kind, float4 additive, f32x4 base)
	{
		if (kind == 0)
		{
			const float4 rotation = kFastMath ? quat_mul_fast(additive, make_float4(base.x, base.y, base.z, base.w)) : quat_mul(additive, make_float4(base.x, base.y, base.z, base.w));
			return f32x4{ rotation.x, rotation.y, rotation.z, rotation.w };
		}
		if (kind == 1)
			return f32x4{ additive.x + base.x, additive.y + base.y, additive.z + base.z, 0.0f };
		if (additive_format == 2)
			return f32x4{ additive.x * base.x, additive.y * base.y, additive.z * base.z, 0.0f };
		return f32x4{ (1.0f + additive.x) * base.x, (1.0f + additive.y) * base.y, (1.0f + additive.z) * base.z, 0.0f };
	}

	// A decoded sub-track of an additive clip goes ONTO the base pose the image already holds
	template<bool kFastMath = false>
	struct additive_image_writer
	{
		f32x4* image;
		uint32_t additive_format;
		__device__ __forceinline__ void operator()(const clip_range_entry& entry, float4 value) const
		{
			const uint32_t quad = entry.quad_index;
			image[quad] = apply_additive_sub_track<kFastMath>(additive_format, quad - entry.track_index * 3u, value, image[quad]);
		}
	};

	// The additive clip applied onto the base pose in `image`, sub-track by sub-track: the constant and default ones from the clip's
	// base pose table (defaults are the track_writer's own: what the pose consumers require), the animated ones as they are decoded.
	template<bool kFastMath = false>
	__device__ __forceinline__ void apply_additive_clip_onto_image(const device_clip& clip, float sample_time, uint32_t rounding_policy, const decode_params& params,
		uint32_t additive_format, uint32_t lane, f32x4* image)
	{
		const uint32_t num_quads = clip.num_tracks * 3u;
		for (uint32_t quad = lane; quad < num_quads; quad += k_wave_size)
		{
			float4 value = load_quad(clip.base_pose, quad);
			const uint32_t marker = __float_as_uint(value.w);
			if (is_special_quad(marker))
			{
				if ((marker & k_quad_animated) != 0)
					continue;
				value.w = (marker & k_quad_default_w_one) != 0 ? 1.0f : 0.0f;
			}
			image[quad] = apply_additive_sub_track<kFastMath>(additive_format, quad - (quad / 3u) * 3u, value, image[quad]);
		}
		decode_animated_into_image<kFastMath>(clip, sample_time, rounding_policy, params, lane, additive_image_writer<kFastMath>{ image, additive_format });
	}

	// ---- blend of K clip instances (aclhip_pose_consumers::num_blend_clips; include/aclhip.h states the operation order) ----------------
	// One image, one wave, like the fused additive path: the first clip is decoded into the image and scaled by its weight, every further
	// clip is accumulated ONTO it sub-track by sub-track -- constant and default sub-tracks from the clip's base pose table, animated ones
	// as they are decoded --, then the rotations are normalized. Rotations are sign aligned with what has been accumulated so far (the
	// sign bias of quat_lerp_no_normalization, math/quatf.h:177-190, against the running sum instead of the first key).
	__device__ __forceinline__ f32x4 blend_accumulate(uint32_t kind, f32x4 accumulated, float4 value, float weight)
	{
		if (kind == 0)
		{
			float dot = accumulated.x * value.x;
			dot = dot + (accumulated.y * value.y);
			dot = dot + (accumulated.z * value.z);
			dot = dot + (accumulated.w * value.w);
			const float signed_weight = dot < 0.0f ? -weight : weight;
			return f32x4{ (value.x * signed_weight) + accumulated.x, (value.y * signed_weight) + accumulated.y, (value.z * signed_weight) + accumulated.z, (value.w * signed_weight) + accumulated.w };
		}
		return f32x4{ (value.x * weight) + accumulated.x, (value.y * weight) + accumulated.y, (value.z * weight) + accumulated.z, 0.0f };
	}

	struct blend_image_writer
	{
		f32x4* image;
		float weight;
		__device__ __forceinline__ void operator()(const clip_range_entry& entry, float4 value) const
		{
			const uint32_t quad = entry.quad_index;
			image[quad] = blend_accumulate(quad - entry.track_index * 3u, image[quad], value, weight);
		}
	};

	// the first clip's pose, complete in `image`, times its weight
	__device__ __forceinline__ void blend_scale_image(f32x4* image, uint32_t num_quads, float weight, uint32_t lane)
	{
		for (uint32_t quad = lane; quad < num_quads; quad += k_wave_size)
		{
			const f32x4 value = image[quad];
			const bool is_rotation = quad % 3u == 0;
			image[quad] = f32x4{ value.x * weight, value.y * weight, value.z * weight, is_rotation ? value.w * weight : 0.0f };
		}
	}

	__device__ __forceinline__ void blend_clip_onto_image(const device_clip& clip, float sample_time, uint32_t rounding_policy, const decode_params& params,
		float weight, uint32_t lane, f32x4* image)
	{
		const uint32_t num_quads = clip.num_tracks * 3u;
		for (uint32_t quad = lane; quad < num_quads; quad += k_wave_size)
		{
			float4 value = load_quad(clip.base_pose, quad);
			const uint32_t marker = __float_as_uint(value.w);
			if (is_special_quad(marker))
			{
				if ((marker & k_quad_animated) != 0)
					continue;
				value.w = (marker & k_quad_default_w_one) != 0 ? 1.0f : 0.0f;
			}
			image[quad] = blend_accumulate(quad - (quad / 3u) * 3u, image[quad], value, weight);
		}
		decode_animated_into_image(clip, sample_time, rounding_policy, params, lane, blend_image_writer{ image, weight });
	}

	__device__ __forceinline__ void blend_normalize_rotations(f32x4* image, uint32_t num_tracks, uint32_t lane)
	{
		for (uint32_t track = lane; track < num_tracks; track += k_wave_size)
		{
			const f32x4 value = image[track * 3u];
			const float4 rotation = quat_normalize(make_float4(value.x, value.y, value.z, value.w));
			image[track * 3u] = f32x4{ rotation.x, rotation.y, rotation.z, rotation.w };
		}
	}

	// May the object space walk normalize this clip's rotations with the short exact forms (aclhip_device.h)? Its quantized rotations must
	// be proven safe (k_clip_short_exact_math); raw ones are any floats, and only a normalizing decode brings those to a norm of 1
	__device__ __forceinline__ uint32_t walk_may_use_short_exact_math(uint32_t clip_flags, uint32_t normalization)
	{
		return (clip_flags & k_clip_short_exact_math) != 0 && ((clip_flags & k_clip_raw_rotations) == 0 || normalization != ACLHIP_NORMALIZE_NEVER) ? 1u : 0u;
	}

	__device__ __forceinline__ void wave_lds_barrier()
	{
		__builtin_amdgcn_s_waitcnt(0);
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
		__builtin_amdgcn_wave_barrier();
		__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
	}

	__device__ __forceinline__ qvv load_qvv(const f32x4* image, uint32_t transform_index)
	{
		const f32x4 r = image[transform_index * 3u + 0], t = image[transform_index * 3u + 1], s = image[transform_index * 3u + 2];
		qvv value;
		value.rotation = make_float4(r.x, r.y, r.z, r.w);
		value.translation = make_float4(t.x, t.y, t.z, 0.0f);
		value.scale = make_float4(s.x, s.y, s.z, 0.0f);
		return value;
	}

	__device__ __forceinline__ void store_qvv(f32x4* image, uint32_t transform_index, const qvv& value)
	{
		image[transform_index * 3u + 0] = f32x4{ value.rotation.x, value.rotation.y, value.rotation.z, value.rotation.w };
		image[transform_index * 3u + 1] = f32x4{ value.translation.x, value.translation.y, value.translation.z, 0.0f };
		image[transform_index * 3u + 2] = f32x4{ value.scale.x, value.scale.y, value.scale.z, 0.0f };
	}

	// Up to 8 instances per workgroup, one wave64 per clip instance to decode: the (additive) clip instance and, when the base is a clip,
	// its base clip instance in a second wave, each into its own LDS image; the two are combined per transform
	// (apply_additive_to_base, core/additive_utils.h:150). local_to_object_space (compression/transform_pose_utils.h:35) is a walk
	// down the hierarchy, parents first, and a depth of a 100 bone skeleton is 4-18 transforms wide: done per wave it would leave
	// most lanes idle for some 135 instructions per depth. So the workgroup's FIRST wave walks all its instances at once, lanes <->
	// (instance, transform of the current step of the schedule aclhip_set_clip_hierarchy made), from copies of the schedules the
	// waves left in LDS next to their poses; then the finished poses stream out. What a caller would otherwise do in further passes over the pose buffer in
	// HBM happens on the image the decode already holds.
	// LDS per instance: [pose image | base image (base clips only) | hierarchy copy (object space only)].
	constexpr uint32_t k_consumer_max_instances = 8;

	// The kernel's static LDS: what the decoding waves leave behind for the walking wave, per instance of the workgroup
	struct consumer_walk_slots
	{
		uint32_t levels[k_consumer_max_instances];				// steps to walk per instance of the workgroup; 0: nothing to do; bit 31: its schedule is NOT in the shared LDS copy
		const uint32_t* schedules[k_consumer_max_instances];	// and the schedule to follow (global memory)
		uint32_t tracks[k_consumer_max_instances];				// transforms of each instance's pose (0: nothing to store)
		uint32_t short_exact[k_consumer_max_instances];			// 1: every clip behind the instance's pose is k_clip_short_exact_math (or the slot has no work)
	};
	constexpr uint32_t k_consumer_max_waves = k_consumer_max_instances * 2;

	// Launch wide facts are template arguments (each instantiation keeps only its own path: registers, code size):
	//   kObjectSpace   local -> object space with the clips' hierarchies
	//   kBase          k_consumer_base_*: no base | a pose buffer in HBM | a clip decoded by a second wave into a second image (the relative
	//                  format) | a clip decoded by the instance's own wave, the additive clip onto it (additive0 / additive1)
	//   kUnitScale     object space without a base while no registered clip has a scale other than 1: the images hold rotation |
	//                  translation only (32 of a transform's 48 bytes)
	//   kMirrored      some registered clip may decode a NEGATIVE scale (or the base is a caller's pose buffer): rtm::qvv_mul's matrix
	//                  route is compiled in (10 more registers: one wave per SIMD less) and the transforms that take it are counted; without it
	//                  none can occur (scales that are sums and products of non negative values).
	constexpr uint32_t k_consumer_base_none = 0, k_consumer_base_buffer = 1, k_consumer_base_second_wave = 2, k_consumer_base_fused = 3;

	template<bool kObjectSpace, uint32_t kBase, bool kUnitScale, bool kMirrored, bool kBlend = false, bool kFast = false>
	__global__ __launch_bounds__(k_consumer_max_waves * k_wave_size) void decompress_poses_consumer_kernel(const device_clip* __restrict__ clips, uint32_t num_clips,
		const uint32_t* __restrict__ clip_ids, const float* __restrict__ sample_times, uint32_t num_instances, decode_params launch_params, consumer_params consumers,
		uint8_t* __restrict__ poses, uint64_t pose_stride_bytes, uint32_t lds_quads_per_image, uint32_t lds_bytes_per_instance, uint32_t packed_block_shape,
		unsigned long long* __restrict__ rejected_count)
	{
		// packed_block_shape: log2 of the instances per workgroup (bits 0..7) | words of LDS reserved for the shared walk schedule (bits 8..31)
		const uint32_t log2_instances_per_block = packed_block_shape & 0xFFu;
		extern __shared__ __attribute__((aligned(16))) uint8_t dynamic_lds[];
		__shared__ consumer_walk_slots walk;		// (what the host subtracts from the LDS it may ask for: host_consumers.inl)
		uint32_t (&walk_levels)[k_consumer_max_instances] = walk.levels;
		const uint32_t* (&walk_schedules)[k_consumer_max_instances] = walk.schedules;
		uint32_t (&walk_tracks)[k_consumer_max_instances] = walk.tracks;
		uint32_t (&walk_short_exact)[k_consumer_max_instances] = walk.short_exact;

		static_assert(!kUnitScale || (kObjectSpace && kBase == k_consumer_base_none), "rotation | translation images: object space without a base");
		static_assert(!kBlend || (!kUnitScale && kBase != k_consumer_base_fused), "a blend accumulates whole qvv images; a base clip is decoded by a second wave");
		static_assert(!kFast || !kBlend, "ACLHIP_CONSUMERS_FAST: not instantiated for blends");
		constexpr bool has_base = kBase != k_consumer_base_none;
		constexpr bool base_is_clip = kBase == k_consumer_base_second_wave || kBase == k_consumer_base_fused;
		// a base clip under additive0 / additive1: ONE wave decodes the base into the instance's image and the additive clip onto it
		// (half the LDS per instance, half the waves: twice the poses a CU holds); otherwise a second wave decodes the base into its own image
		constexpr bool fused_base = kBase == k_consumer_base_fused;
		constexpr bool two_waves = kBase == k_consumer_base_second_wave;
		constexpr bool object_space = kObjectSpace;
		constexpr bool unit_scale = kUnitScale;
		ACLHIP_PHASE_STAMP(0);

		// wave -> (instance slot of the workgroup, role): role 1 waves (base clips only) decode the slot's base
		const uint32_t lane = threadIdx.x & (k_wave_size - 1);
		const uint32_t wave_in_block = __builtin_amdgcn_readfirstlane(threadIdx.x / k_wave_size);
		const uint32_t slot = wave_in_block & ((1u << log2_instances_per_block) - 1u);
		const uint32_t role = wave_in_block >> log2_instances_per_block;
		const uint32_t waves_per_instance = two_waves ? 2u : 1u;
		const uint32_t instance = (blockIdx.x << log2_instances_per_block) + slot;

		uint8_t* instance_lds = dynamic_lds + size_t(slot) * lds_bytes_per_instance;
		f32x4* image = reinterpret_cast<f32x4*>(instance_lds);
		f32x4* base_image = image + lds_quads_per_image;
		// one LDS copy of the walk schedule per workgroup, behind the instances' images: the instances of a workgroup usually share
		// a skeleton (identical hierarchies are one image, see aclhip_set_clip_hierarchy), and every word kept per instance costs residency
		uint32_t* shared_schedule = reinterpret_cast<uint32_t*>(dynamic_lds + (size_t(lds_bytes_per_instance) << log2_instances_per_block));
		const uint32_t* schedule = nullptr;

		uint32_t num_tracks = 0;		// stays 0 for a wave without work: past the batch, refused instance, empty track list
		uint32_t num_levels = 0;
		// the walk's normalize may take the short exact forms when every rotation it meets comes out of clips that are proven safe for
		// them (norms near 1; a caller's base pose buffer holds anything)
		uint32_t short_exact = kBase == k_consumer_base_buffer ? 0u : 1u;
		if (instance < num_instances)
		{
			const uint32_t clip_id = as_constant(clip_ids)[instance];
			// (every field in registers of its own: load_clip_fields, kernels_pose.inl)
			const device_clip clip = load_clip_fields(clips, clip_id < num_clips ? clip_id : 0);

			// refused: unknown / scalar clips, object space without a hierarchy, poses larger than the launch's LDS images, bases that
			// are unknown or describe another number of transforms (the reference asserts matching track counts where it combines them).
			// Both waves of an instance come to the same verdict; the first one reports it.
			// A launch is shaped for its batch when it is enqueued (launch_consumers: LDS image sizes from the pose stride, kernel
			// instantiation from what the registry holds) and meets its clips when it runs: a pose that does not fit the stride or the
			// image, a scaled clip under the rotation | translation images, a clip that may decode a negative scale in a launch
			// compiled without rtm::qvv_mul's matrix route -- registered behind a captured launch's back -- are refused here, not decoded wrongly.
			constexpr bool multiplies_transforms = object_space || kBase == k_consumer_base_buffer || kBase == k_consumer_base_second_wave;
			bool refused = clip_id >= num_clips || !is_transform_clip(clip.flags) || (object_space && clip.hierarchy == nullptr)
				// (the image holds no more transforms than a pose row: launch_consumers sizes it from the stride -- one test for both)
				|| clip.num_tracks * (unit_scale ? 2u : 3u) > lds_quads_per_image || (unit_scale && (clip.flags & k_clip_scaled) != 0)
				|| (kBase == k_consumer_base_buffer && uint64_t(clip.num_tracks) * 48u > consumers.base_pose_stride_bytes)
				|| (!kMirrored && multiplies_transforms && !base_is_clip && (clip.flags & k_clip_negative_scale) != 0);

			const uint32_t rounding_policy = __builtin_amdgcn_readfirstlane(instance_rounding_policy_of(launch_params, instance));
			// the instance's own looping policy (decompress.h:149) goes for every clip decoded on its behalf -- its base, its blend partners
			decode_params params = launch_params;
			params.looping_policy = uint8_t(__builtin_amdgcn_readfirstlane(instance_looping_policy_of(launch_params, instance)));

			short_exact &= walk_may_use_short_exact_math(clip.flags, params.normalization);
			device_clip base_clip = clip;
			if (base_is_clip)
			{
				const uint32_t base_clip_id = as_constant(consumers.base_clip_ids)[instance];
				base_clip = load_clip_fields(clips, base_clip_id < num_clips ? base_clip_id : 0);
				refused = refused || base_clip_id >= num_clips || !is_transform_clip(base_clip.flags) || base_clip.num_tracks != clip.num_tracks
					|| (!kMirrored && multiplies_transforms && ((clip.flags | base_clip.flags) & k_clip_negative_scale) != 0);
				short_exact &= walk_may_use_short_exact_math(base_clip.flags, params.normalization);
				if (!refused && two_waves && role == 1 && clip.num_tracks != 0)
					decode_pose_into_image<kFast>(base_clip, as_constant(consumers.base_sample_times)[instance], rounding_policy, params, lane, base_image);
			}

			if (kBlend && !refused)
			{
				// every clip of the blend: known, a transform clip, as many tracks as the first
				for (uint32_t k = 1; k < consumers.num_blend_clips; ++k)
				{
					const uint32_t blend_clip_id = as_constant(consumers.blend_clip_ids)[size_t(instance) * (consumers.num_blend_clips - 1u) + (k - 1u)];
					const ACLHIP_CONSTANT device_clip* record = as_constant(clips) + (blend_clip_id < num_clips ? blend_clip_id : 0);
					refused = refused || blend_clip_id >= num_clips || !is_transform_clip(record->flags) || record->num_tracks != clip.num_tracks
						|| (!kMirrored && multiplies_transforms && (record->flags & k_clip_negative_scale) != 0);
					short_exact &= walk_may_use_short_exact_math(record->flags, params.normalization);
				}
			}

			if (refused)
			{
				if (lane == 0 && role == 0)
					atomicAdd(rejected_count, 1ull);
			}
			else if (clip.num_tracks != 0)
			{
				num_tracks = clip.num_tracks;
				if (role == 0)
				{
					if (object_space)
					{
						// The walk schedule for this many instances per workgroup, requested BEFORE the decode (until round 4 behind it: three
						// more dependent round trips -- offset, header, words -- at the end of every wave's chain, 1.7 of a decode's 6.3 us).
						// One scalar load for the schedule's header (aclhip_set_clip_hierarchy: {offset, steps, words, 0} per workgroup size,
						// in flight next to the seek's sample records), then the words travel global -> LDS by DMA while the pose is decoded:
						//     num_steps | words | step_end[num_steps] | transform | parent << 16 in step order, padded to whole 16 byte pieces
						// Every wave leaves its schedule in the shared copy: the same words when they share it (the copy is only used then).
						// A schedule longer than the launch reserved LDS for (a hierarchy set behind a captured launch's back) stays in
						// global memory and the walk reads it there.
						const u32x4 header = ((const ACLHIP_CONSTANT u32x4*)clip.hierarchy)[log2_instances_per_block];
						schedule = clip.hierarchy + header.x;
						num_levels = header.y;
						const uint32_t num_words = header.z;
						if (num_words <= (packed_block_shape >> 8))
						{
							for (uint32_t base = 0; base < num_words; base += k_wave_size * 4u)
								if (base + lane * 4u < num_words)
									__builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(schedule + base + lane * 4u),
										(__attribute__((address_space(3))) void*)(shared_schedule + base), 16, 0, 0);
						}
						else
							num_levels |= 0x80000000u;
					}
					if (fused_base)
					{
						decode_pose_into_image<kFast>(base_clip, as_constant(consumers.base_sample_times)[instance], rounding_policy, params, lane, image);
						wave_lds_barrier();		// the base pose is complete (its DMA has landed)
						apply_additive_clip_onto_image<kFast>(clip, as_constant(sample_times)[instance], rounding_policy, params, consumers.additive_format, lane, image);
					}
					else if (unit_scale)
						decode_unit_scale_pose_into_image<kFast>(clip, as_constant(sample_times)[instance], rounding_policy, params, lane, image);
					else
						decode_pose_into_image<kFast>(clip, as_constant(sample_times)[instance], rounding_policy, params, lane, image);
					if constexpr (kBlend)
					{
						const uint32_t num_blend_clips = consumers.num_blend_clips;
						const ACLHIP_CONSTANT float* weights = as_constant(consumers.blend_weights) + size_t(instance) * num_blend_clips;
						wave_lds_barrier();		// the first pose is complete (its DMA has landed)
						blend_scale_image(image, clip.num_tracks * 3u, weights[0], lane);
						for (uint32_t k = 1; k < num_blend_clips; ++k)
						{
							const size_t entry = size_t(instance) * (num_blend_clips - 1u) + (k - 1u);
							const device_clip blend_clip = load_clip_fields(clips, as_constant(consumers.blend_clip_ids)[entry]);
							wave_lds_barrier();		// every quad has its sum so far
							blend_clip_onto_image(blend_clip, as_constant(consumers.blend_sample_times)[entry], rounding_policy, params, weights[k], lane, image);
						}
						wave_lds_barrier();
						blend_normalize_rotations(image, clip.num_tracks, lane);
					}
				}
			}
		}

		// both images of every instance are complete
		if (two_waves)
			__syncthreads();
		else
			wave_lds_barrier();

		if (has_base && !fused_base)
		{
			const f32x4* base_source = base_is_clip ? base_image : reinterpret_cast<const f32x4*>(consumers.base_poses + uint64_t(instance) * consumers.base_pose_stride_bytes);
			for (uint32_t transform_index = role * k_wave_size + lane; transform_index < num_tracks; transform_index += waves_per_instance * k_wave_size)
			{
				const qvv additive = load_qvv(image, transform_index);
				const qvv base = load_qvv(base_source, transform_index);
				store_qvv(image, transform_index, apply_additive_to_base<kMirrored>(consumers.additive_format, base, additive));
				// additive_clip_format8::relative is a qvv_mul (core/additive_utils.h:128-160)
				if constexpr (kMirrored)
				{
					const uint64_t mirrored = __ballot(consumers.additive_format == 1 && qvv_mul_takes_matrix_path(additive, base));
					if (mirrored != 0 && lane == uint32_t(__builtin_ctzll(mirrored)))
						atomicAdd(rejected_count + 1, (unsigned long long)__builtin_popcountll(mirrored));
				}
			}
		}

		if (object_space)
		{
			if (lane == 0 && role == 0)
			{
				walk_levels[slot] = num_levels;
				walk_schedules[slot] = schedule;
				walk_tracks[slot] = num_tracks;
				walk_short_exact[slot] = num_tracks != 0 ? short_exact : 1u;
			}
			__syncthreads();
			ACLHIP_PHASE_STAMP(1);

			// ONE wave walks and then stores the workgroup's poses; the others are done and give their wave slots and registers back (a
			// pose waits in LDS for the walk about as long as its decode took: with every wave parked at a barrier the wave slots, not the
			// LDS, decided how many poses a CU holds). The walking wave rotates with the workgroup index: waves land on SIMDs by their
			// index inside the workgroup, and walks that all ran on a CU's first SIMD would queue there.
			if (wave_in_block != (blockIdx.x & ((blockDim.x / k_wave_size) - 1u)))
				return;
			{
				// lanes <-> (instance slot, transform of the current step): slot = lane % instances, lane / instances picks the slot's
				// transform inside the step. A transform's parent was scheduled in an earlier step: final by the time it is read.
				const uint32_t walk_slot = lane & ((1u << log2_instances_per_block) - 1u);
				const uint32_t first = lane >> log2_instances_per_block;
				f32x4* slot_image = reinterpret_cast<f32x4*>(dynamic_lds + size_t(walk_slot) * lds_bytes_per_instance);
				const uint32_t slot_steps = walk_levels[walk_slot] & 0x7FFFFFFFu;
				const bool slot_schedule_is_shared = (walk_levels[walk_slot] & 0x80000000u) == 0;
				const uint32_t* slot_schedule = walk_schedules[walk_slot];

				const auto walk = [&](const auto* schedule_words, auto scale_is_one, auto short_exact_tag)
				{
					constexpr bool k_unit_scale = decltype(scale_is_one)::value;
					constexpr bool k_short_exact = decltype(short_exact_tag)::value;		// sqrt_rn_short / rcp_rn_short in the normalize (aclhip_device.h)
					const auto* pairs = schedule_words + 2u + slot_steps;
					uint32_t step_start = 0;
					for (uint32_t step = 0; __any(int(step < slot_steps)) != 0; ++step)
					{
						if (step < slot_steps)
						{
							const uint32_t step_end = schedule_words[2 + step];
							const uint32_t pair_index = step_start + first;
							if (pair_index < step_end)
							{
								const uint32_t pair = pairs[pair_index];		// transform | parent << 16
								if constexpr (k_unit_scale)
								{
									// rotation | translation images; qvv_mul with both scales 1: translation * 1 is the translation itself
									const uint32_t child_quad = (pair & 0xFFFFu) * 2u, parent_quad = (pair >> 16) * 2u;
									const f32x4 child_rotation = slot_image[child_quad], child_translation = slot_image[child_quad + 1];
									const f32x4 parent_rotation = slot_image[parent_quad], parent_translation = slot_image[parent_quad + 1];
									const float4 parent_quat = make_float4(parent_rotation.x, parent_rotation.y, parent_rotation.z, parent_rotation.w);
									const float4 child_quat = make_float4(child_rotation.x, child_rotation.y, child_rotation.z, child_rotation.w);
									const float4 child_vector = make_float4(child_translation.x, child_translation.y, child_translation.z, 0.0f);
									const float4 rotation = kFast ? quat_normalize_fast(quat_mul_fast(child_quat, parent_quat)) : quat_normalize<k_short_exact>(quat_mul(child_quat, parent_quat));
									const float4 rotated = kFast ? quat_mul_vector3_fast(child_vector, parent_quat) : quat_mul_vector3(child_vector, parent_quat);
									slot_image[child_quad] = f32x4{ rotation.x, rotation.y, rotation.z, rotation.w };
									slot_image[child_quad + 1] = f32x4{ rotated.x + parent_translation.x, rotated.y + parent_translation.y, rotated.z + parent_translation.z, 0.0f };
								}
								else
								{
									const qvv child = load_qvv(slot_image, pair & 0xFFFFu), parent = load_qvv(slot_image, pair >> 16);
									qvv object;
									if constexpr (kMirrored)
									{
										const uint64_t mirrored = __ballot(qvv_mul_takes_matrix_path(child, parent));
										if (mirrored != 0 && lane == uint32_t(__builtin_ctzll(mirrored)))
											atomicAdd(rejected_count + 1, (unsigned long long)__builtin_popcountll(mirrored));
										object = kFast ? qvv_mul_fast(child, parent) : qvv_mul(child, parent);
										if (mirrored != 0 && qvv_mul_takes_matrix_path(child, parent))
											object = qvv_mul_through_matrices(child, parent);
									}
									else
									{
										// no registered clip can decode a negative scale and the base is a clip: products and sums of non
										// negative scales -- nothing to count, nothing to route
										object = kFast ? qvv_mul_fast(child, parent) : qvv_mul(child, parent);
									}
									object.rotation = kFast ? quat_normalize_fast(object.rotation) : quat_normalize<k_short_exact>(object.rotation);
									store_qvv(slot_image, pair & 0xFFFFu, object);
								}
							}
							step_start = step_end;
						}
						wave_lds_barrier();
					}
				};

				// all instances that walk follow the same schedule? then the shared LDS copy is theirs; otherwise each reads its own
				// from global memory (rare: mixed skeletons inside one workgroup)
				// the rest of the workgroup waits for this wave: it goes first on its SIMD
				__builtin_amdgcn_s_setprio(3);
				const uint64_t walkers = __ballot(slot_steps != 0);
				if (walkers != 0)
				{
					const uint32_t leader = uint32_t(__builtin_ctzll(walkers));
					const uint64_t mine = reinterpret_cast<uint64_t>(slot_schedule);
					const uint64_t first_schedule = (uint64_t(__shfl(uint32_t(mine >> 32), int(leader))) << 32) | __shfl(uint32_t(mine), int(leader));
					const bool shared_copy = __all(int(slot_steps == 0 || (mine == first_schedule && slot_schedule_is_shared))) != 0;
					const bool short_exact_walk = !kFast && __all(int(walk_short_exact[walk_slot] != 0)) != 0;
					const auto walk_with = [&](auto scale_is_one, auto short_exact_tag)
					{
						if (shared_copy)
							walk(static_cast<const uint32_t*>(shared_schedule), scale_is_one, short_exact_tag);
						else
							walk(as_constant(slot_schedule), scale_is_one, short_exact_tag);
					};
					if (short_exact_walk)
						walk_with(std::integral_constant<bool, unit_scale>(), std::true_type());
					else
						walk_with(std::integral_constant<bool, unit_scale>(), std::false_type());
				}
				__builtin_amdgcn_s_setprio(0);
			}
			wave_lds_barrier();
			ACLHIP_PHASE_STAMP(2);

			const uint32_t instances_per_block = 1u << log2_instances_per_block;
			for (uint32_t store_slot = 0; store_slot < instances_per_block; ++store_slot)
			{
				const uint32_t slot_quads = walk_tracks[store_slot] * 3u;
				const f32x4* slot_image = reinterpret_cast<const f32x4*>(dynamic_lds + size_t(store_slot) * lds_bytes_per_instance);
				f32x4* slot_pose = reinterpret_cast<f32x4*>(poses + uint64_t((blockIdx.x << log2_instances_per_block) + store_slot) * pose_stride_bytes);
				if (unit_scale)
				{
					// rotation | translation in LDS, rotation | translation | scale (1, 1, 1) in the pose
					for (uint32_t quad = lane; quad < slot_quads; quad += k_wave_size)
					{
						const uint32_t track = quad / 3u;
						const uint32_t kind = quad - track * 3u;
						const f32x4 value = kind == 2 ? f32x4{ 1.0f, 1.0f, 1.0f, 0.0f } : slot_image[track * 2u + min(kind, 1u)];
						store_streaming(&slot_pose[quad], value);
					}
				}
				else
					for (uint32_t quad = lane; quad < slot_quads; quad += k_wave_size)
						store_streaming(&slot_pose[quad], slot_image[quad]);
			}
			ACLHIP_PHASE_STAMP(3);
			return;
		}
		else if (two_waves)
			__syncthreads();
		else
			wave_lds_barrier();

		const uint32_t num_quads = num_tracks * 3u;
		f32x4* pose = reinterpret_cast<f32x4*>(poses + uint64_t(instance) * pose_stride_bytes);
		for (uint32_t quad = role * k_wave_size + lane; quad < num_quads; quad += waves_per_instance * k_wave_size)
			store_streaming(&pose[quad], image[quad]);
		ACLHIP_PHASE_STAMP(3);
	}
