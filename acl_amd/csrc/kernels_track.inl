// kernels_track.inl -- part of aclhip.hip (one translation unit; included there, in this order, not compiled on its own).
// decompress_track_kernel: single (instance, bone) requests -- decompress_track_v0 (decompression/impl/decompression.transform.h:1753-2050).
//
// One wave64 takes 64 consecutive requests and works on them TOGETHER (round 5; until then one lane did one request start to end:
// 41 vector loads per wave, most of them issued for a handful of active lanes, and three 16 byte stores 48 bytes apart per lane --
// WRITE_SIZE 2.0 x the transforms, the texture unit 80 % busy, 226.6 us for 4 M requests, profiles/r04_track_requests_pmc_*.txt):
//   1. lanes <-> requests: clip handle, sample time, track index in; when every request of the wave names the SAME clip -- a crowd of one
//      rig, the bones of one character -- the clip record and everything derived from it live on the scalar unit (two s_loads instead of
//      eight vector loads per lane); a wave of mixed clips gathers the 64 byte heads of its records four lanes per record and hands them
//      out through LDS (gather_clip_records, round 6). The seek is per lane either way (sample times
//      differ): two 16 byte sample records. The track's three base pose quads tell what each sub-track is: constant (the value itself),
//      default, or animated (marker + ordinal).
//   2. the wave's ANIMATED (request, kind) pairs -- about 0.4 per request on CMU-shaped clips, not 3 -- are counted with three ballots and
//      packed into consecutive lanes (all rotations, then translations, then scales): lanes <-> animated sub-tracks, 64 per pass, usually
//      ONE pass per wave instead of three sparse loops. A decode lane picks its request's seek result out of LDS, fetches its plan rows
//      (one when both keys sit in one segment) and clip range, reads each key's bits with one aligned 16 byte load, and runs the same
//      unpack / range / W / lerp / normalize as the pose kernels (aclhip_device.h: bit for bit the same values).
//   3. decoded and constant quads meet in a 3 KiB LDS image of the wave's 64 transforms, which leaves as three 1 KiB contiguous
//      streaming stores. Requests whose output is partly or wholly withheld (an unknown clip, a bad track index, default sub-tracks in
//      the "skipped" mode, the tail of the batch) send their wave down the per lane store path instead.
// The reference sums the widths of every preceding animated sub-track to find a track's bits (skip_*_groups +
// count_animated_group_bit_size, animated_track_cache.transform.h:1105-1192,1664-1707): O(track index). The registration time plan holds
// that prefix sum.

	// what a decode lane needs to know about the request its sub-track belongs to: 32 bytes in LDS, as two 16 byte parts [part][request]
	// (the interpolation alpha comes over ds_bpermute, the clip's tables from the scalar clip record or -- waves of mixed clips -- as ONE
	// pointer per request, the clip's clip range table, over ds_bpermute as well: the plan sits directly in front of it in the clip's
	// allocation (host_clips.inl), so the rows are kept as distances back from there. Until round 6 the decode lane fetched its request's
	// clip record again for the two pointers: one more dependent load in every pass of a mixed wave. 5 KiB of LDS per wave all told, so that registers, not LDS, decide how many waves a
	// CU holds. Measured, 4 M requests on one clip: 64 bytes of state per request and the list beside the image (7.75 KiB, 20 waves per
	// CU) 74.3 us; this form at 70 registers = 7 waves per SIMD 64.3 us; squeezed into 64 registers for 8 (15 spilled) 85.4 us.)
	struct track_request_state
	{
		const uint8_t* data[2];					// first stored keyframe of each key's data source (seek_state::animated_track_data)
		uint32_t rows[2];						// plan entries from the first entry of each key's segment to the END of the plan = the clip range table
												// ((segments - segment index) x animated sub-tracks); bit 31 of rows[0]: the clip's k_clip_short_exact_math
		uint32_t bit_offsets[2];				// seek_state::key_frame_bit_offsets
	};
	static_assert(sizeof(track_request_state) == 32, "two 16 byte parts");
	constexpr uint32_t k_track_row_short_exact_math = 0x80000000u;

#if !defined(ACLHIP_TRACK_WAVES_PER_EU)
	#define ACLHIP_TRACK_WAVES_PER_EU 7
#endif
#if defined(ACLHIP_TRACK_NARROW_KEYS)
	constexpr bool k_track_wide_key_loads = false;
#else
	constexpr bool k_track_wide_key_loads = true;
#endif
	constexpr uint32_t k_track_image_bytes = k_wave_size * 48;				// 64 transforms; before that the list of the wave's animated (request, kind) pairs, 4 bytes each
	constexpr uint32_t k_track_state_bytes = k_wave_size * 32;
	constexpr uint32_t k_track_lds_bytes_per_wave = k_track_image_bytes + k_track_state_bytes;
	static_assert(k_wave_size * 3 * 4 <= k_track_image_bytes, "the list fits where the image will be");

	__device__ __forceinline__ void track_wave_barrier()
	{
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
		__builtin_amdgcn_wave_barrier();
		__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
	}

	// A wave of mixed clips: every lane's own clip record. Left to itself a lane fetches the seven 16 byte pieces of its record it uses --
	// seven load instructions that touch 64 different cache lines each, and the texture unit takes a line per cycle: 450 of the wave's
	// ~1 500 cycles there (round 6; 4 M requests of 256 clips as drawn: 198 us against 77 us for one clip). Instead FOUR lanes fetch the
	// 64 byte head of one record (k_clip_head_bytes: everything a seek and a track request read from it) -- four load instructions over
	// 16 lines each, 64 contiguous bytes per line -- and the records change hands in LDS (4 KiB of the wave's 5, before the list and the
	// image use them). The second half of a record -- database tiers, the table defaults' bind pose -- is fetched per lane, and only by a
	// wave that has such a request.
	__device__ __forceinline__ device_clip gather_clip_records(const device_clip* clips, uint32_t num_clips, uint32_t clip_id, bool known_clip, uint32_t lane, uint8_t* wave_lds, bool wants_tail)
	{
		static_assert(k_clip_head_bytes == 64 && k_track_image_bytes + k_track_state_bytes >= k_wave_size * k_clip_head_bytes, "64 heads fit in the wave's LDS");
		u32x4* staging = reinterpret_cast<u32x4*>(wave_lds);			// [request][piece]
		const uint32_t safe_id = known_clip ? clip_id : 0u;			// (entry 0 always exists; its bytes are not used)
		const uint32_t piece = lane & 3u;
		#pragma unroll
		for (uint32_t round = 0; round < 4; ++round)
		{
			const uint32_t request = round * 16u + (lane >> 2);
			const uint32_t id = uint32_t(__builtin_amdgcn_ds_bpermute(int(request * 4u), int(safe_id)));
			// (piece p of request r in slot (p + r / 4) % 4 of the request's four: the lanes that read a piece back -- 64 bytes apart --
			// then start on different banks)
			staging[request * 4u + ((piece + (request >> 2)) & 3u)] = ((const ACLHIP_CONSTANT u32x4*)(clips + id))[piece];
		}
		track_wave_barrier();
		u32x4 raw[8];
		#pragma unroll
		for (uint32_t i = 0; i < 4; ++i)
			raw[i] = staging[lane * 4u + ((i + (lane >> 2)) & 3u)];
		const u32x4 zero = { 0u, 0u, 0u, 0u };
		#pragma unroll
		for (uint32_t i = 4; i < 8; ++i)
			raw[i] = zero;
		track_wave_barrier();		// the staging area is the list's and the image's from here on
		device_clip clip;
		__builtin_memcpy(&clip, raw, sizeof(clip));
		// the tail: a clip bound to a database reads its tiers, a table default the clip's bind pose (wave uniform)
		if (wants_tail || __builtin_amdgcn_ballot_w64(known_clip && (clip.flags & k_clip_has_database) != 0) != 0)
		{
			const ACLHIP_CONSTANT u32x4* source = (const ACLHIP_CONSTANT u32x4*)(clips + safe_id);
			#pragma unroll
			for (uint32_t i = 4; i < 8; ++i)
				raw[i] = source[i];
			__builtin_memcpy(&clip, raw, sizeof(clip));
		}
		return clip;
	}

	// Step 1 for one lane's request against its clip record (in SGPRs when the wave shares the clip, in VGPRs otherwise): validation,
	// seek, the track's three base pose quads. Returns false for a request that is refused (decompression.transform.h:1766-1768).
	// out_animated: bit k = sub-track kind k is animated (out_ordinals[k] = its ordinal); the other kinds have their final value in
	// out_quads[k] and whether it is stored at all in out_store[k].
	__device__ __forceinline__ bool prepare_track_request(const device_clip& clip, float sample_time, uint32_t track_index, uint32_t rounding_policy, uint32_t looping_policy,
		const uint8_t* track_rounding_policies, const decode_params& params, track_request_state& out_state, float& out_lerp_alpha, float4 (&out_quads)[3], bool (&out_store)[3], uint32_t& out_animated, uint32_t (&out_ordinals)[3])
	{
		out_animated = 0;
		// invalid track index (decompression.transform.h:1766-1768); an empty clip lands here as well
		if (!is_transform_clip(clip.flags) || track_index >= clip.num_tracks)
			return false;

		seek_state state;
		seek(clip, sample_time, rounding_policy, looping_policy, state);

		// decompress_track_v0 folds a per track policy into the alpha and always interpolates (decompression.transform.h:1975-1983)
		float lerp_alpha = state.interpolation_alpha;
		if (params.per_track_rounding != 0)
		{
			uint32_t policy = rounding_policy;
			if (rounding_policy == k_round_per_track)
				policy = track_rounding_policies != nullptr ? track_rounding_policies[track_index] : k_round_none;
			lerp_alpha = apply_rounding_policy(lerp_alpha, policy);
		}
		out_lerp_alpha = lerp_alpha;

		out_state.data[0] = state.animated_track_data[0];
		out_state.data[1] = state.animated_track_data[1];
		out_state.rows[0] = ((clip.num_segments - state.segment_index[0]) * clip.num_animated) | ((clip.flags & k_clip_short_exact_math) != 0 ? k_track_row_short_exact_math : 0u);
		out_state.rows[1] = (clip.num_segments - state.segment_index[1]) * clip.num_animated;
		out_state.bit_offsets[0] = state.key_frame_bit_offsets[0];
		out_state.bit_offsets[1] = state.key_frame_bit_offsets[1];

		// base pose quads: constant (real W), animated (marker + ordinal) or default (marker)
		float4 quads[3];
		#pragma unroll
		for (uint32_t kind = 0; kind < 3; ++kind)
			quads[kind] = load_quad(clip.base_pose, track_index * 3u + kind);

		#pragma unroll
		for (uint32_t kind = 0; kind < 3; ++kind)
		{
			float4 value = quads[kind];
			const uint32_t marker = __float_as_uint(value.w);
			bool store = true;
			if (is_special_quad(marker) && (marker & k_quad_animated) != 0)
			{
				out_animated |= 1u << kind;
				out_ordinals[kind] = marker & k_quad_ordinal_mask;
			}
			else if (is_special_quad(marker))
				value = resolve_quad(params, value, track_index * 3u + kind, store, bind_pose_of(clip));
			else if (params.normalization == ACLHIP_NORMALIZE_ALWAYS && kind == 0 && (clip.flags & k_clip_full_rotations) == 0)
				value = quat_normalize(value);		// constant_track_cache.transform.h:163-175
			out_quads[kind] = value;
			out_store[kind] = store;
		}
		return true;
	}

	template<uint32_t kFastMath>
	__device__ __forceinline__ void decompress_track_requests(const device_clip* __restrict__ clips, uint32_t num_clips,
		const uint32_t* __restrict__ clip_ids, const float* __restrict__ sample_times, const uint32_t* __restrict__ track_indices, uint32_t num_instances,
		const decode_params& params, float4* __restrict__ transforms, unsigned long long* __restrict__ rejected_count)
	{
		__shared__ __attribute__((aligned(16))) uint8_t track_lds[k_waves_per_block * k_track_lds_bytes_per_wave];

		const uint32_t lane = threadIdx.x & (k_wave_size - 1);
		const uint32_t wave_in_block = __builtin_amdgcn_readfirstlane(threadIdx.x / k_wave_size);
		const uint32_t first_instance = (blockIdx.x * k_waves_per_block + wave_in_block) * k_wave_size;		// wave uniform
		if (first_instance >= num_instances)
			return;
		const uint32_t instance = first_instance + lane;
		const bool in_batch = instance < num_instances;

		uint8_t* wave_lds = track_lds + wave_in_block * k_track_lds_bytes_per_wave;
		f32x4* image = reinterpret_cast<f32x4*>(wave_lds);												// [request * 3 + kind]
		uint32_t* list = reinterpret_cast<uint32_t*>(wave_lds);											// (read into registers before the image takes its place)
		u32x4* state_parts = reinterpret_cast<u32x4*>(wave_lds + k_track_image_bytes);					// [part][request]

		// ---- 1. lanes <-> requests ---------------------------------------------------------------------------------------------------
		const uint32_t clamped_instance = in_batch ? instance : first_instance;
		const uint32_t clip_id = clip_ids[clamped_instance];
		const float sample_time = sample_times[clamped_instance];
		const uint32_t track_index = track_indices[clamped_instance];
		const uint32_t rounding_policy = instance_rounding_policy_of(params, clamped_instance);
		const uint32_t looping_policy = instance_looping_policy_of(params, clamped_instance);
		const uint8_t* track_rounding_policies = params.per_track_rounding != 0 ? instance_track_rounding_of(params, clamped_instance) : nullptr;

		const bool known_clip = in_batch && clip_id < num_clips;
		const uint32_t first_clip_id = __builtin_amdgcn_readfirstlane(clip_id);			// lane 0 is always in the batch
		const bool shared_clip = __builtin_amdgcn_ballot_w64(in_batch && clip_id != first_clip_id) == 0 && first_clip_id < num_clips;	// wave uniform

		track_request_state state;
		float lerp_alpha = 0.0f;
		float4 quads[3] = { make_float4(0.0f, 0.0f, 0.0f, 0.0f), make_float4(0.0f, 0.0f, 0.0f, 0.0f), make_float4(0.0f, 0.0f, 0.0f, 0.0f) };		// (refused requests write these to LDS, never to memory)
		bool store[3] = { false, false, false };
		uint32_t animated = 0;
		uint32_t ordinals[3] = { 0, 0, 0 };
		bool accepted = false;
		// the clip range table of the request's clip (the plan ends where it starts): the shared clip's for the decode lanes (wave uniform),
		// or every lane's own
		const clip_range_entry* shared_clip_ranges = nullptr;
		const clip_range_entry* own_clip_ranges = nullptr;
		if (shared_clip)
		{
			const device_clip clip = load_clip(clips, first_clip_id);
			shared_clip_ranges = clip.clip_ranges;
			if (in_batch)
				accepted = prepare_track_request(clip, sample_time, track_index, rounding_policy, looping_policy, track_rounding_policies, params, state, lerp_alpha, quads, store, animated, ordinals);
		}
		else
		{
			const device_clip clip = gather_clip_records(clips, num_clips, clip_id, known_clip, lane, wave_lds, params.user_defaults != 0);
			own_clip_ranges = clip.clip_ranges;
			if (known_clip)
				accepted = prepare_track_request(clip, sample_time, track_index, rounding_policy, looping_policy, track_rounding_policies, params, state, lerp_alpha, quads, store, animated, ordinals);
		}

		// refused requests are counted, one atomic per wave
		const uint64_t refused = __builtin_amdgcn_ballot_w64(in_batch && !accepted);
		if (refused != 0 && lane == 0)
			atomicAdd(rejected_count, (unsigned long long)__builtin_popcountll(refused));

		// ---- 2. lanes <-> the wave's animated sub-tracks ------------------------------------------------------------------------------
		const uint64_t animated_lanes[3] = { __builtin_amdgcn_ballot_w64((animated & 1u) != 0), __builtin_amdgcn_ballot_w64((animated & 2u) != 0), __builtin_amdgcn_ballot_w64((animated & 4u) != 0) };
		const uint32_t num_animated[3] = { uint32_t(__builtin_popcountll(animated_lanes[0])), uint32_t(__builtin_popcountll(animated_lanes[1])), uint32_t(__builtin_popcountll(animated_lanes[2])) };
		const uint32_t total_animated = num_animated[0] + num_animated[1] + num_animated[2];		// wave uniform, at most 192: three passes

		uint32_t entries[3] = { 0, 0, 0 };
		if (total_animated != 0)
		{
			uint32_t first_position = 0;
			#pragma unroll
			for (uint32_t kind = 0; kind < 3; ++kind)
			{
				if ((animated >> kind) & 1u)
				{
					const uint32_t position = first_position + __builtin_amdgcn_mbcnt_hi(uint32_t(animated_lanes[kind] >> 32), __builtin_amdgcn_mbcnt_lo(uint32_t(animated_lanes[kind]), 0u));
					list[position] = lane | (kind << 6) | (ordinals[kind] << 8);
				}
				first_position += num_animated[kind];
			}
			if (animated != 0)
			{
				u32x4 parts[2];
				__builtin_memcpy(parts, &state, sizeof(state));
				state_parts[lane] = parts[0];
				state_parts[k_wave_size + lane] = parts[1];
			}
			track_wave_barrier();
			entries[0] = list[min(lane, total_animated - 1)];
			if (total_animated > k_wave_size)
			{
				entries[1] = list[min(k_wave_size + lane, total_animated - 1)];
				entries[2] = list[min(2 * k_wave_size + lane, total_animated - 1)];
			}
			track_wave_barrier();		// the list is in registers: its place is the image's now
		}

		// what is not animated is final already
		#pragma unroll
		for (uint32_t kind = 0; kind < 3; ++kind)
			if (((animated >> kind) & 1u) == 0)
				image[lane * 3u + kind] = f32x4{ quads[kind].x, quads[kind].y, quads[kind].z, quads[kind].w };

		#pragma unroll 1
		for (uint32_t pass = 0; pass * k_wave_size < total_animated; ++pass)
		{
			const bool valid = pass * k_wave_size + lane < total_animated;
			const uint32_t entry = pass == 0 ? entries[0] : (pass == 1 ? entries[1] : entries[2]);
			const uint32_t source_lane = entry & 63u;
			const uint32_t kind = (entry >> 6) & 3u;
			const uint32_t ordinal = entry >> 8;

			u32x4 parts[2] = { state_parts[source_lane], state_parts[k_wave_size + source_lane] };
			track_request_state request;
			__builtin_memcpy(&request, parts, sizeof(request));
			const float request_alpha = __uint_as_float(uint32_t(__builtin_amdgcn_ds_bpermute(int(source_lane * 4u), int(__float_as_uint(lerp_alpha)))));

			// the clip's tables: the wave's one clip, or -- a wave of mixed clips -- each request's own, from the lane that prepared it
			const clip_range_entry* clip_ranges = shared_clip_ranges;
			if (!shared_clip)
			{
				const uint64_t own = reinterpret_cast<uint64_t>(own_clip_ranges);
				const uint32_t low = uint32_t(__builtin_amdgcn_ds_bpermute(int(source_lane * 4u), int(uint32_t(own))));
				const uint32_t high = uint32_t(__builtin_amdgcn_ds_bpermute(int(source_lane * 4u), int(uint32_t(own >> 32))));
				clip_ranges = reinterpret_cast<const clip_range_entry*>((uint64_t(high) << 32) | low);
			}

			// (plan_entry and clip_range_entry are both 32 bytes: the plan's last entry is clip_ranges[-1])
			static_assert(sizeof(plan_entry) == sizeof(clip_range_entry), "the plan is addressed from the clip range table");
			const plan_entry* plan_end = reinterpret_cast<const plan_entry*>(clip_ranges);
			const uint32_t row0 = request.rows[0] & ~k_track_row_short_exact_math, row1 = request.rows[1];
			const plan_entry plan0 = load_entry(plan_end - row0, ordinal);
			const plan_entry plan1 = row1 == row0 ? plan0 : load_entry(plan_end - row1, ordinal);
			const clip_range_entry clip_range = load_entry(clip_ranges, ordinal);

			seek_state key_state;
			key_state.animated_track_data[0] = request.data[0];
			key_state.animated_track_data[1] = request.data[1];
			key_state.segment_index[0] = key_state.segment_index[1] = 0;		// (the rows are resolved already)
			key_state.key_frame_bit_offsets[0] = request.bit_offsets[0];
			key_state.key_frame_bit_offsets[1] = request.bit_offsets[1];
			key_state.interpolation_alpha = request_alpha;
			key_state.uses_single_segment = false;

			// the raw bit rate is rare: only a wave that actually meets one pays for its code path
			const bool has_raw = __any(int(is_raw_width(plan0.bit_offset_and_width >> 24) || is_raw_width(plan1.bit_offset_and_width >> 24))) != 0;
			const bool short_exact_math = (request.rows[0] & k_track_row_short_exact_math) != 0;
			float4 value;
			if (!has_raw)
				value = decode_animated_sub_track<false, false, k_track_wide_key_loads, kFastMath>(key_state, plan0, plan1, clip_range, kind == 0, k_round_none, request_alpha, params.normalization, false, short_exact_math);
			else
				value = decode_animated_sub_track<true, false, k_track_wide_key_loads>(key_state, plan0, plan1, clip_range, kind == 0, k_round_none, request_alpha, params.normalization, false, false);
			if (valid)
				image[source_lane * 3u + kind] = f32x4{ value.x, value.y, value.z, value.w };
		}

		track_wave_barrier();

		// ---- 3. the wave's 64 transforms leave -----------------------------------------------------------------------------------------
		const bool stores_all = accepted && store[0] && store[1] && store[2];
		if (__builtin_amdgcn_ballot_w64(stores_all) == ~0ull)
		{
			// three 1 KiB contiguous stores
			f32x4 staged[3];
			#pragma unroll
			for (uint32_t row = 0; row < 3; ++row)
				staged[row] = image[row * k_wave_size + lane];
			float4* out = transforms + size_t(first_instance) * 3u + lane;
			#pragma unroll
			for (uint32_t row = 0; row < 3; ++row)
				store_streaming(out + row * k_wave_size, staged[row]);
			return;
		}

		// a request that is refused, or whose default sub-tracks are skipped, leaves (part of) its transform as the caller had it
		if (accepted)
		{
			#pragma unroll
			for (uint32_t kind = 0; kind < 3; ++kind)
				if (store[kind])
					store_streaming(&transforms[size_t(instance) * 3u + kind], image[lane * 3u + kind]);
		}
	}

	__global__ __launch_bounds__(k_block_size) __attribute__((amdgpu_waves_per_eu(ACLHIP_TRACK_WAVES_PER_EU, ACLHIP_TRACK_WAVES_PER_EU))) void decompress_track_kernel(const device_clip* __restrict__ clips, uint32_t num_clips,
		const uint32_t* __restrict__ clip_ids, const float* __restrict__ sample_times, const uint32_t* __restrict__ track_indices, uint32_t num_instances,
		decode_params params, float4* __restrict__ transforms, unsigned long long* __restrict__ rejected_count)
	{
		decompress_track_requests<0>(clips, num_clips, clip_ids, sample_times, track_indices, num_instances, params, transforms, rejected_count);
	}

	// (ACLHIP_DECODE_FAST: single track requests are served by the kernel above. A variant with the hardware's 1 ulp square root and
	// fused multiply-adds existed for half of round 6: the arithmetic it saves is 18 of the 460 vector instructions of a wave, and it
	// never measured faster -- 79.3 against 76.7 us on 4 M requests, 107.6 against 78.2 us once the record gather took its registers
	// (24 spilled at 7 waves per SIMD; 86.1 us at 6) -- profiles/r06_experiments.md 4.)
