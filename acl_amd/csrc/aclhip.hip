// aclhip.hip -- gfx950 decode kernels and the host side of the C ABI of libaclhip.so (include/aclhip.h).
//
// Kernels (device helpers in aclhip_device.h):
//   decompress_tracks_kernel / decompress_tracks_any_settings_kernel
//       one wave64 per (clip instance, window of 320 pose quads): scalar seek, base pose DMA'd global -> LDS, lanes <-> animated
//       sub-tracks decode in place into the LDS image (bit unpack, segment + clip range, W, lerp, normalize), the window streams
//       out 1 KiB per store instruction. The second entry point is the same body with per track rounding, the non default
//       default sub-track modes and always-normalize compiled in.
//   decompress_poses_consumer_kernel   the same decode, whole pose per wave, followed by what callers do next with a local pose -- apply
//                                      an additive clip onto its base, local -> object space -- before the pose leaves LDS.
//   decompress_track_kernel            one thread per (instance, bone) request; the registration time plan replaces the
//                                      reference's O(track index) skip over preceding widths.
//   decompress_scalar_tracks_kernel    scalar track lists: one wave64 per (instance, 256 tracks), lanes <-> tracks.
//   decompress_scalar_track_kernel     scalar track lists: one thread per (instance, track) request.
//   apply_tier_metadata_kernel         publishes / retires database tier metadata behind a stream ordered bulk copy.
//
// Every output store is a streaming store (store_streaming, aclhip_device.h): poses must not evict clip data from the L2s.
//
// Host side: context and clip / database registries (clips live in shared HBM slabs), blob validation, registration time tables,
// walk schedules of hierarchies, locality order of instance lists, launches.
//
// Build (acl_amd/build.py): hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -mllvm -amdgpu-kernarg-preload-count=16 aclhip.hip -ldl -o ../lib/libaclhip.so
#include <hip/hip_runtime.h>
#include <dlfcn.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <unordered_set>
#include <type_traits>
#include <vector>

#include "../../include/aclhip.h"
#include "acl_format.h"
#include "aclhip_device.h"

namespace aclhip
{
	constexpr uint32_t k_wave_size = 64;
#if !defined(ACLHIP_WAVES_PER_BLOCK)
	#define ACLHIP_WAVES_PER_BLOCK 4
#endif
	constexpr uint32_t k_waves_per_block = ACLHIP_WAVES_PER_BLOCK;
	constexpr uint32_t k_block_size = k_wave_size * k_waves_per_block;

	// Value of a default sub-track (unpack_default_*_sub_tracks, decompression.transform.h:575-675,883-985,1203-1310, and the
	// "no scale" loop :1653-1680). `identity` is the track_writer default for the kind (identity / zero / legacy scale).
	__device__ __forceinline__ float4 default_quad(const decode_params& params, uint32_t kind, uint32_t track_index, float4 identity, bool& out_store)
	{
		const uint32_t mode = params.default_modes[kind];
		out_store = mode != ACLHIP_DEFAULT_SKIPPED;

		if (params.default_values != nullptr && (mode == ACLHIP_DEFAULT_CONSTANT || mode == ACLHIP_DEFAULT_VARIABLE))
		{
			const float* src = params.default_values + (mode == ACLHIP_DEFAULT_VARIABLE ? size_t(track_index) * 12 : 0) + kind * 4;
			return make_float4(src[0], src[1], src[2], kind == 0 ? src[3] : 0.0f);
		}

		if (kind == 2 && mode != ACLHIP_DEFAULT_LEGACY)
			return make_float4(1.0f, 1.0f, 1.0f, 0.0f);		// track_writer::get_constant_default_scale (core/track_writer.h:169)

		return identity;
	}

	typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
	typedef uint32_t u32x16 __attribute__((ext_vector_type(16)));
	typedef float f32x4 __attribute__((ext_vector_type(4)));

	// The whole 128 byte clip record in two scalar loads (wave uniform address)
	__device__ __forceinline__ device_clip load_clip(const device_clip* clips, uint32_t clip_id)
	{
		const ACLHIP_CONSTANT u32x16* source = (const ACLHIP_CONSTANT u32x16*)(clips + clip_id);
		struct { u32x16 lo, hi; } raw = { source[0], source[1] };
		device_clip clip;
		__builtin_memcpy(&clip, &raw, sizeof(clip));
		return clip;
	}

	// A 32 byte table entry (plan_entry / clip_range_entry) in two 16 byte loads
	template<class entry_t>
	__device__ __forceinline__ entry_t load_entry(const entry_t* table, uint32_t index)
	{
		static_assert(sizeof(entry_t) == 32, "two dwordx4 loads");
		const ACLHIP_CONSTANT u32x4* source = (const ACLHIP_CONSTANT u32x4*)(table + index);
		struct { u32x4 lo, hi; } raw = { source[0], source[1] };
		entry_t entry;
		__builtin_memcpy(&entry, &raw, sizeof(entry));
		return entry;
	}

	__device__ __forceinline__ float4 load_quad(const float4* table, uint32_t index)
	{
		const f32x4 raw = ((const ACLHIP_CONSTANT f32x4*)table)[index];
		return make_float4(raw.x, raw.y, raw.z, raw.w);
	}

	// What the any-settings pose kernel stores for a quad of the LDS image: default sub-tracks -- still tagged in their W lane, every
	// other quad holds a real W >= +0 by now -- follow the default sub-track modes, the rest passes through.
	__device__ __forceinline__ float4 resolve_quad(const decode_params& params, float4 value, uint32_t quad, bool& out_store)
	{
		out_store = true;
		const uint32_t marker = __float_as_uint(value.w);
		if (int32_t(marker) >= 0)
			return value;

		const uint32_t track_index = quad / 3u;
		const uint32_t kind = quad - track_index * 3u;
		value.w = (marker & k_quad_default_w_one) != 0 ? 1.0f : 0.0f;
		return default_quad(params, kind, track_index, value, out_store);
	}

	// Lanes <-> the animated sub-tracks [first_ordinal, end_ordinal) of one pose window, decoded into their quads of the window's LDS
	// image (image[0] = quad first_quad of the pose). Most sample times fall between two keyframes of ONE segment: both keys then
	// share a plan row and it is fetched once (a third less table traffic through the texture unit).
	template<bool kSingleSegment, bool kPolicies>
	__device__ __forceinline__ void decode_window_sub_tracks_with(const clip_range_entry* __restrict__ clip_ranges, const seek_state& state, const decode_params& params,
		uint32_t rounding_policy, uint32_t normalization, uint32_t first_ordinal, uint32_t end_ordinal, uint32_t first_quad, uint32_t lane, f32x4* image)
	{
		// kPolicies: per track rounding (and the sample normalization it implies)
		const bool normalize_samples = kPolicies && normalization == ACLHIP_NORMALIZE_ALWAYS;

		for (uint32_t animated_ordinal = first_ordinal + lane; animated_ordinal < end_ordinal; animated_ordinal += k_wave_size)
		{
			const plan_entry plan0 = load_entry(state.plan[0], animated_ordinal);
			const plan_entry plan1_loaded = kSingleSegment ? plan0 : load_entry(state.plan[1], animated_ordinal);
			const plan_entry& plan1 = kSingleSegment ? plan0 : plan1_loaded;
			const clip_range_entry clip_range = load_entry(clip_ranges, animated_ordinal);
			const bool is_rotation = is_rotation_entry(clip_range);

			uint32_t policy = k_round_none;
			if (kPolicies)
			{
				// track_writer::get_rounding_policy (core/track_writer.h:97)
				policy = rounding_policy;
				if (rounding_policy == k_round_per_track)
					policy = params.track_rounding_policies != nullptr ? params.track_rounding_policies[clip_range.track_index] : k_round_none;
			}

			// the raw bit rate is rare: only a wave that actually meets one (in these two segments) pays for its code path
			const bool has_raw = __any(int((plan0.bit_offset_and_width >> 24) == 32u || (plan1.bit_offset_and_width >> 24) == 32u)) != 0;

			float4 value;
			if (!has_raw)
				value = decode_animated_sub_track<false, kPolicies>(state, plan0, plan1, clip_range, is_rotation, policy, state.interpolation_alpha, normalization, normalize_samples);
			else
				value = decode_animated_sub_track<true, kPolicies>(state, plan0, plan1, clip_range, is_rotation, policy, state.interpolation_alpha, normalization, normalize_samples);

			// a decoded W is never negative (a square root, or +0): the marker the base pose carried in this quad is gone
			const f32x4 packed = { value.x, value.y, value.z, value.w };
			image[clip_range.quad_index - first_quad] = packed;
		}
	}

	template<bool kAnySettings>
	__device__ __forceinline__ void decode_window_sub_tracks(const clip_range_entry* __restrict__ clip_ranges, const seek_state& state, const decode_params& params,
		uint32_t rounding_policy, uint32_t normalization, uint32_t first_ordinal, uint32_t end_ordinal, uint32_t first_quad, uint32_t lane, f32x4* image)
	{
		if (kAnySettings && params.per_track_rounding != 0)
		{
			if (state.uses_single_segment)
				decode_window_sub_tracks_with<true, true>(clip_ranges, state, params, rounding_policy, normalization, first_ordinal, end_ordinal, first_quad, lane, image);
			else
				decode_window_sub_tracks_with<false, true>(clip_ranges, state, params, rounding_policy, normalization, first_ordinal, end_ordinal, first_quad, lane, image);
		}
		else if (state.uses_single_segment)
			decode_window_sub_tracks_with<true, false>(clip_ranges, state, params, rounding_policy, normalization, first_ordinal, end_ordinal, first_quad, lane, image);
		else
			decode_window_sub_tracks_with<false, false>(clip_ranges, state, params, rounding_policy, normalization, first_ordinal, end_ordinal, first_quad, lane, image);
	}

	// The pose kernels. One wave64 per (instance, pose window): a window is k_image_chunk_quads consecutive quads of the pose (a
	// 100 bone pose is a single window), built in 5 KiB of LDS:
	//   1. the scalar prologue finds the clip and seeks (4 dependent scalar loads);
	//   2. meanwhile the window's slice of the clip's base pose is DMA'd global -> LDS (global_load_lds, no VGPRs);
	//   3. lanes <-> the animated sub-tracks that land in the window (a contiguous range of ordinals: the tables are ordered by
	//      window) decode straight into their quad of the LDS image;
	//   4. the finished window streams out, 16 bytes per lane, 1 KiB of contiguous HBM per store instruction.
	// Windows of one pose go to consecutive waves: each repeats the (scalar) seek, none waits for another, and the chain of
	// dependent memory round trips per wave stays as short as for a small pose.
	//
	// kAnySettings = false is the common case -- track_writer defaults, no per track rounding, normalization != always: the DMA source
	// is the clip's RESOLVED pose (defaults written out) and step 4 is a plain copy. kAnySettings = true takes every settings
	// combination: the DMA source is the marker tagged base pose, the decode honours per track rounding, and step 4 resolves what
	// is not animated (default sub-track modes, caller supplied defaults, always-normalize).
	template<bool kAnySettings>
	__device__ __forceinline__ void decompress_tracks_window(const device_clip* __restrict__ clips, uint32_t num_clips,
		const uint32_t* __restrict__ clip_ids, const float* __restrict__ sample_times, uint32_t num_instances, uint32_t windows_per_instance,
		const decode_params& params, uint8_t* __restrict__ poses, uint64_t pose_stride_bytes, uint32_t lds_quads_per_wave,
		unsigned long long* __restrict__ rejected_count)
	{
		extern __shared__ __attribute__((aligned(16))) uint8_t dynamic_lds[];

		const uint32_t lane = threadIdx.x & (k_wave_size - 1);
		const uint32_t wave_in_block = __builtin_amdgcn_readfirstlane(threadIdx.x / k_wave_size);
		const uint32_t work_item = blockIdx.x * k_waves_per_block + wave_in_block;
		uint32_t instance = work_item;
		uint32_t window = 0;
		if (windows_per_instance != 1)
		{
			instance = work_item / windows_per_instance;
			window = work_item - instance * windows_per_instance;
		}
		if (instance >= num_instances)
			return;

		// wave uniform prologue on the scalar unit: instance -> clip record -> sample records
		const uint32_t clip_id = as_constant(clip_ids)[instance];
		const float sample_time = as_constant(sample_times)[instance];
		const device_clip clip = load_clip(clips, clip_id < num_clips ? clip_id : 0);
		if (clip_id >= num_clips || !is_transform_clip(clip.flags))
		{
			if (lane == 0 && window == 0)
				atomicAdd(rejected_count, 1ull);
			return;
		}

		// an empty track list (decompression.transform.h:1531-1533) or a pose that ends before this window
		const uint32_t num_quads = clip.num_tracks * 3u;
		const uint32_t first_quad = window * k_image_chunk_quads;
		if (first_quad >= num_quads)
			return;
		const uint32_t window_quads = min(num_quads - first_quad, k_image_chunk_quads);

		// the window's animated sub-tracks: image_chunks[window] .. image_chunks[window + 1]
		uint32_t first_ordinal = 0, end_ordinal = clip.num_animated;
		if (num_quads > k_image_chunk_quads)
		{
			first_ordinal = as_constant(clip.image_chunks)[window];
			end_ordinal = as_constant(clip.image_chunks)[window + 1];
		}

		f32x4* image = reinterpret_cast<f32x4*>(dynamic_lds) + size_t(wave_in_block) * lds_quads_per_wave;

		// With the track_writer's own default sub-track modes the resolved pose already holds what default sub-tracks decode to; any
		// other mode starts from the tagged base pose and resolves the tags when the window is stored
		const bool resolve_defaults = kAnySettings && params.standard_default_modes == 0;

		// base pose window -> LDS image, asynchronously: lane i of pass p fetches quad first + p * 64 + i into image[p * 64 + i]
		{
			const ACLHIP_CONSTANT f32x4* source = (const ACLHIP_CONSTANT f32x4*)(resolve_defaults ? clip.base_pose : clip.resolved_pose) + first_quad;
			for (uint32_t base = 0; base < window_quads; base += k_wave_size)
			{
				if (base + lane < window_quads)
					__builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(source + base + lane),
						(__attribute__((address_space(3))) void*)(image + base), 16, 0, 0);
			}
		}

		const uint32_t rounding_policy = params.instance_rounding_policies != nullptr
			? __builtin_amdgcn_readfirstlane(uint32_t(params.instance_rounding_policies[instance]))
			: uint32_t(params.rounding_policy);
		const uint32_t normalization = params.normalization;

		seek_state state;
		seek(clip, sample_time, rounding_policy, params.looping_policy, state);

		if (kAnySettings && normalization == ACLHIP_NORMALIZE_ALWAYS)
		{
			// rotation_normalization_policy_t::always also normalizes CONSTANT rotations (constant_track_cache.transform.h:163-175):
			// done in the image before the animated sub-tracks (normalized by their decode) replace their markers. Rare (debug
			// settings): this path waits for the base pose instead of overlapping it with the decode.
			__builtin_amdgcn_s_waitcnt(0);
			__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
			__builtin_amdgcn_wave_barrier();
			__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
			for (uint32_t quad = lane; quad < window_quads; quad += k_wave_size)
			{
				// (slots of animated rotations hold a tag, or zeros in the resolved pose: whatever this makes of them is overwritten)
				const f32x4 value = image[quad];
				if ((first_quad + quad) % 3u == 0 && int32_t(__float_as_uint(value.w)) >= 0)
				{
					const float4 normalized = quat_normalize(make_float4(value.x, value.y, value.z, value.w));
					image[quad] = f32x4{ normalized.x, normalized.y, normalized.z, normalized.w };
				}
			}
		}

		// lanes <-> animated sub-tracks of this window
		decode_window_sub_tracks<kAnySettings>(clip.clip_ranges, state, params, rounding_policy, normalization, first_ordinal, end_ordinal, first_quad, lane, image);

		// DMA and the wave's own LDS writes must have landed before lanes read each other's quads
		__builtin_amdgcn_s_waitcnt(0);
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
		__builtin_amdgcn_wave_barrier();
		__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

		// LDS -> registers -> HBM: the whole window is read first, then the stores go out back to back from one base address with
		// immediate offsets; full 1 KiB rows take no per lane predicate, only the last (partial) row does
		constexpr uint32_t k_rows = k_image_chunk_quads / k_wave_size;
		const uint32_t full_rows = window_quads / k_wave_size;			// wave uniform
		f32x4 staged[k_rows];
		#pragma unroll
		for (uint32_t r = 0; r < k_rows; ++r)
			staged[r] = image[min(r * k_wave_size + lane, lds_quads_per_wave - 1)];

		// (the row is read here, not in the prologue: one SGPR pair less across the decode)
		const uint32_t row = params.instance_rows != nullptr ? as_constant(params.instance_rows)[instance] : instance;
		f32x4* pose = reinterpret_cast<f32x4*>(poses + uint64_t(row) * pose_stride_bytes) + first_quad + lane;

		// any-settings: rows are 64 quads apart and 64 % 3 == 1, so a lane's sub-track kind advances by one per row
		const uint32_t lane_quad = first_quad + lane;
		const uint32_t lane_track = lane_quad / 3u;
		uint32_t kind = lane_quad - lane_track * 3u;
		const bool user_defaults = kAnySettings && params.default_values != nullptr;		// wave uniform

		#pragma unroll
		for (uint32_t r = 0; r < k_rows; ++r)
		{
			bool store = r < full_rows || (r == full_rows && r * k_wave_size + lane < window_quads);
			f32x4 value = staged[r];
			if (kAnySettings && resolve_defaults)
			{
				// default sub-tracks still carry their tag in the W lane (every other quad holds a real W >= +0 by now) and follow the
				// default sub-track modes (unpack_default_*_sub_tracks, decompression.transform.h:575-675,883-985,1203-1310,1653-1680)
				const uint32_t marker = __float_as_uint(value.w);
				const bool is_default = int32_t(marker) < 0;
				const uint32_t mode = kind == 0 ? params.default_modes[0] : (kind == 1 ? params.default_modes[1] : params.default_modes[2]);
				store = store && !(is_default && mode == ACLHIP_DEFAULT_SKIPPED);
				if (is_default)
				{
					// the image holds the track_writer default's xyz (identity / zero / the clip's legacy default scale)
					value.w = (marker & k_quad_default_w_one) != 0 ? 1.0f : 0.0f;
					if (kind == 2 && mode != ACLHIP_DEFAULT_LEGACY)
						value = f32x4{ 1.0f, 1.0f, 1.0f, 0.0f };		// track_writer::get_constant_default_scale (core/track_writer.h:169)
				}
				if (user_defaults && is_default && (mode == ACLHIP_DEFAULT_CONSTANT || mode == ACLHIP_DEFAULT_VARIABLE))
				{
					const uint32_t track_index = (lane_quad + r * k_wave_size) / 3u;
					const float* source = params.default_values + (mode == ACLHIP_DEFAULT_VARIABLE ? size_t(track_index) * 12 : 0) + kind * 4;
					value = f32x4{ source[0], source[1], source[2], kind == 0 ? source[3] : 0.0f };
				}
				kind = kind == 2 ? 0u : kind + 1u;
			}
			if (store)
				store_streaming(&pose[r * k_wave_size], value);
		}
	}

	__global__ __launch_bounds__(k_block_size) void decompress_tracks_kernel(const device_clip* __restrict__ clips, uint32_t num_clips,
		const uint32_t* __restrict__ clip_ids, const float* __restrict__ sample_times, uint32_t num_instances, uint32_t windows_per_instance,
		decode_params params, uint8_t* __restrict__ poses, uint64_t pose_stride_bytes, uint32_t lds_quads_per_wave, unsigned long long* __restrict__ rejected_count)
	{
		decompress_tracks_window<false>(clips, num_clips, clip_ids, sample_times, num_instances, windows_per_instance, params, poses, pose_stride_bytes, lds_quads_per_wave, rejected_count);
	}

	// 8 waves per SIMD (64 VGPRs) matter more to this variant than the few instructions the allocator saves with 65
	__global__ __launch_bounds__(k_block_size) __attribute__((amdgpu_waves_per_eu(8, 8))) void decompress_tracks_any_settings_kernel(const device_clip* __restrict__ clips, uint32_t num_clips,
		const uint32_t* __restrict__ clip_ids, const float* __restrict__ sample_times, uint32_t num_instances, uint32_t windows_per_instance,
		decode_params params, uint8_t* __restrict__ poses, uint64_t pose_stride_bytes, uint32_t lds_quads_per_wave, unsigned long long* __restrict__ rejected_count)
	{
		decompress_tracks_window<true>(clips, num_clips, clip_ids, sample_times, num_instances, windows_per_instance, params, poses, pose_stride_bytes, lds_quads_per_wave, rejected_count);
	}

	// ---- pose consumers -----------------------------------------------------------------------------------------------
	// Decodes the whole local pose of one clip instance into an LDS image (image[0] = quad 0), window by window like the pose kernels
	// but in ONE wave, because what follows needs every transform of the pose. Common-case settings only (see launch_consumers).
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

		seek_state state;
		seek(clip, sample_time, rounding_policy, params.looping_policy, state);

		if (num_quads <= k_image_chunk_quads)
			decode_window_sub_tracks<false>(clip.clip_ranges, state, params, rounding_policy, params.normalization, 0, clip.num_animated, 0, lane, image);
		else
		{
			const uint32_t num_windows = (num_quads + k_image_chunk_quads - 1) / k_image_chunk_quads;
			for (uint32_t window = 0; window < num_windows; ++window)
			{
				const uint32_t first_ordinal = as_constant(clip.image_chunks)[window];
				const uint32_t end_ordinal = as_constant(clip.image_chunks)[window + 1];
				decode_window_sub_tracks<false>(clip.clip_ranges, state, params, rounding_policy, params.normalization, first_ordinal, end_ordinal, 0, lane, image);
			}
		}
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
	constexpr uint32_t k_consumer_max_waves = k_consumer_max_instances * 2;

	__global__ __launch_bounds__(k_consumer_max_waves * k_wave_size) void decompress_poses_consumer_kernel(const device_clip* __restrict__ clips, uint32_t num_clips,
		const uint32_t* __restrict__ clip_ids, const float* __restrict__ sample_times, uint32_t num_instances, decode_params params, consumer_params consumers,
		uint8_t* __restrict__ poses, uint64_t pose_stride_bytes, uint32_t lds_quads_per_image, uint32_t lds_bytes_per_instance, uint32_t log2_instances_per_block,
		unsigned long long* __restrict__ rejected_count)
	{
		extern __shared__ __attribute__((aligned(16))) uint8_t dynamic_lds[];
		__shared__ uint32_t walk_levels[k_consumer_max_instances];				// steps to walk per instance of the workgroup; 0: nothing to do
		__shared__ const uint32_t* walk_schedules[k_consumer_max_instances];	// and the schedule to follow (global memory)

		const bool has_base = consumers.additive_format != 0;
		const bool base_is_clip = has_base && consumers.base_clip_ids != nullptr;
		const bool object_space = consumers.object_space != 0;

		// wave -> (instance slot of the workgroup, role): role 1 waves (base clips only) decode the slot's base
		const uint32_t lane = threadIdx.x & (k_wave_size - 1);
		const uint32_t wave_in_block = __builtin_amdgcn_readfirstlane(threadIdx.x / k_wave_size);
		const uint32_t slot = wave_in_block & ((1u << log2_instances_per_block) - 1u);
		const uint32_t role = wave_in_block >> log2_instances_per_block;
		const uint32_t waves_per_instance = base_is_clip ? 2u : 1u;
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
		if (instance < num_instances)
		{
			const uint32_t clip_id = as_constant(clip_ids)[instance];
			const device_clip clip = load_clip(clips, clip_id < num_clips ? clip_id : 0);

			// refused: unknown / scalar clips, object space without a hierarchy, poses larger than the launch's LDS images, bases that
			// are unknown or describe another number of transforms (the reference asserts matching track counts where it combines them).
			// Both waves of an instance come to the same verdict; the first one reports it.
			bool refused = clip_id >= num_clips || !is_transform_clip(clip.flags) || (object_space && clip.hierarchy == nullptr) || clip.num_tracks * 3u > lds_quads_per_image;

			const uint32_t rounding_policy = params.instance_rounding_policies != nullptr
				? __builtin_amdgcn_readfirstlane(uint32_t(params.instance_rounding_policies[instance]))
				: uint32_t(params.rounding_policy);

			if (base_is_clip)
			{
				const uint32_t base_clip_id = as_constant(consumers.base_clip_ids)[instance];
				const device_clip base_clip = load_clip(clips, base_clip_id < num_clips ? base_clip_id : 0);
				refused = refused || base_clip_id >= num_clips || !is_transform_clip(base_clip.flags) || base_clip.num_tracks != clip.num_tracks;
				if (!refused && role == 1 && clip.num_tracks != 0)
					decode_pose_into_image(base_clip, as_constant(consumers.base_sample_times)[instance], rounding_policy, params, lane, base_image);
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
					decode_pose_into_image(clip, as_constant(sample_times)[instance], rounding_policy, params, lane, image);
					if (object_space)
					{
						// the walk schedule for this many instances per workgroup (see aclhip_set_clip_hierarchy):
						// num_steps | words | step_end[num_steps] | transform | parent << 16 in step order
						schedule = clip.hierarchy + as_constant(clip.hierarchy)[log2_instances_per_block];
						num_levels = as_constant(schedule)[0];
						// every wave leaves its schedule in the shared copy: the same words when they share it (the copy is only used then)
						const uint32_t num_words = as_constant(schedule)[1];
						for (uint32_t word = lane; word < num_words; word += k_wave_size)
							shared_schedule[word] = schedule[word];
					}
				}
			}
		}

		// both images of every instance are complete
		if (base_is_clip)
			__syncthreads();
		else
			wave_lds_barrier();

		if (has_base)
		{
			const f32x4* base_source = base_is_clip ? base_image : reinterpret_cast<const f32x4*>(consumers.base_poses + uint64_t(instance) * consumers.base_pose_stride_bytes);
			for (uint32_t transform_index = role * k_wave_size + lane; transform_index < num_tracks; transform_index += waves_per_instance * k_wave_size)
			{
				const qvv additive = load_qvv(image, transform_index);
				const qvv base = load_qvv(base_source, transform_index);
				store_qvv(image, transform_index, apply_additive_to_base(consumers.additive_format, base, additive));
			}
		}

		if (object_space)
		{
			if (lane == 0 && role == 0)
			{
				walk_levels[slot] = num_levels;
				walk_schedules[slot] = schedule;
			}
			__syncthreads();

			// the walking wave rotates with the workgroup index: waves land on SIMDs by their index inside the workgroup, and walks that
			// all ran on a CU's first SIMD would queue there
			if (wave_in_block == (blockIdx.x & ((blockDim.x / k_wave_size) - 1u)))
			{
				// lanes <-> (instance slot, transform of the current step): slot = lane % instances, lane / instances picks the slot's
				// transform inside the step. A transform's parent was scheduled in an earlier step: final by the time it is read.
				const uint32_t walk_slot = lane & ((1u << log2_instances_per_block) - 1u);
				const uint32_t first = lane >> log2_instances_per_block;
				f32x4* slot_image = reinterpret_cast<f32x4*>(dynamic_lds + size_t(walk_slot) * lds_bytes_per_instance);
				const uint32_t slot_steps = walk_levels[walk_slot];
				const uint32_t* slot_schedule = walk_schedules[walk_slot];

				const auto walk = [&](const auto* schedule_words)
				{
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
								qvv object = qvv_mul(load_qvv(slot_image, pair & 0xFFFFu), load_qvv(slot_image, pair >> 16));
								object.rotation = quat_normalize(object.rotation);
								store_qvv(slot_image, pair & 0xFFFFu, object);
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
					if (__all(int(slot_steps == 0 || mine == first_schedule)) != 0)
						walk(static_cast<const uint32_t*>(shared_schedule));
					else
						walk(as_constant(slot_schedule));
				}
				__builtin_amdgcn_s_setprio(0);
			}
			__syncthreads();
		}
		else if (base_is_clip)
			__syncthreads();
		else
			wave_lds_barrier();

		const uint32_t num_quads = num_tracks * 3u;
		f32x4* pose = reinterpret_cast<f32x4*>(poses + uint64_t(instance) * pose_stride_bytes);
		for (uint32_t quad = role * k_wave_size + lane; quad < num_quads; quad += waves_per_instance * k_wave_size)
			store_streaming(&pose[quad], image[quad]);
	}

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

	// One track of a scalar track list, C components: unpack both key frames, expand, lerp, store C packed floats.
	// The two table rows of one scalar track: header (bit offset | width, 1 / max) and range row (min[C], extent[C])
	template<uint32_t C>
	struct scalar_track_tables
	{
		scalar_track_header header;
		float range[2 * C];
	};

	template<uint32_t C>
	__device__ __forceinline__ scalar_track_tables<C> load_scalar_track_tables(const scalar_track_header* headers, const float* ranges, uint32_t track_index)
	{
		// rows are only as aligned as their size allows (8 / 16 / 24 / 32 bytes from a 16 byte aligned base): dword aligned vector loads
		typedef float range_row __attribute__((ext_vector_type(2 * C == 6 ? 8 : 2 * C), aligned(4)));
		typedef float range_quad __attribute__((ext_vector_type(4), aligned(4)));
		// read only tables: constant address space loads may be hoisted above the stores of a previous track
		typedef uint32_t header_words __attribute__((ext_vector_type(2)));
		scalar_track_tables<C> tables;
		const header_words header_raw = *(const ACLHIP_CONSTANT header_words*)(headers + track_index);
		tables.header.bit_offset_and_width = header_raw.x;
		tables.header.inv_max_value = __uint_as_float(header_raw.y);
		const ACLHIP_CONSTANT float* row_address = as_constant(ranges) + size_t(track_index) * 2 * C;
		if constexpr (C == 3)
		{
			const range_quad lo = *(const ACLHIP_CONSTANT range_quad*)row_address;		// no 6 wide vector type: 4 + 1 + 1
			const float hi0 = row_address[4], hi1 = row_address[5];
			tables.range[0] = lo.x; tables.range[1] = lo.y; tables.range[2] = lo.z; tables.range[3] = lo.w; tables.range[4] = hi0; tables.range[5] = hi1;
		}
		else
		{
			const range_row row = *(const ACLHIP_CONSTANT range_row*)row_address;
			#pragma unroll
			for (uint32_t c = 0; c < 2 * C; ++c)
				tables.range[c] = row[c];
		}
		return tables;
	}

	// Where the bits of the two key frames come from: global memory (the blob), or the wave's LDS copy of both frames
	struct scalar_frames
	{
		const uint8_t* blob;				// global path
		const uint32_t* lds_frame[2];		// LDS path: dwords of each frame's copy ...
		uint32_t lds_bit_base[2];			// ... and the blob relative bit address of its first dword
		uint32_t frame_bit_offset[2];		// key frame * bits per frame
	};

	// One track of a scalar track list, C components: unpack both key frames, expand, lerp, store C packed floats.
	template<uint32_t C, bool kFromLds>
	__device__ __forceinline__ void decode_scalar_track(const scalar_frames& frames, const scalar_track_tables<C>& tables, float alpha, float* destination)
	{
		const scalar_track_header& header = tables.header;
		const float* range = tables.range;
		const uint32_t num_bits = header.bit_offset_and_width >> 24;
		const uint32_t track_bit_offset = header.bit_offset_and_width & 0x00FFFFFFu;
		const ACLHIP_CONSTANT uint8_t* animated_values = as_constant(frames.blob);

		// Straight line code for the common case, every lane whatever its width: a constant track (width 0) reads a harmless window
		// at bit 0 of the blob, extracts a zero wide field and is put right by the final select; the raw width (32) is rare and, on
		// the global path, only a wave that meets one pays for its 64 bit windows.
		const bool is_constant = num_bits == 0;
		const bool is_raw = num_bits == 32;
		const uint32_t field_bits = is_raw ? 0u : num_bits;
		const bool wave_has_raw = !kFromLds && __any(int(is_raw)) != 0;

		float value[C];
		#pragma unroll
		for (uint32_t c = 0; c < C; ++c)
		{
			const uint32_t offset0 = frames.frame_bit_offset[0] + track_bit_offset + c * num_bits;
			const uint32_t offset1 = frames.frame_bit_offset[1] + track_bit_offset + c * num_bits;

			float value0, value1;
			if constexpr (kFromLds)
			{
				// two aligned dwords hold any field of up to 32 bits: big endian 64 bit window, shifted to the field's first bit
				const uint32_t bit0 = is_constant ? 0u : offset0 - frames.lds_bit_base[0];
				const uint32_t bit1 = is_constant ? 0u : offset1 - frames.lds_bit_base[1];
				const uint32_t* words0 = frames.lds_frame[0] + (bit0 >> 5);
				const uint32_t* words1 = frames.lds_frame[1] + (bit1 >> 5);
				const uint64_t window0 = ((uint64_t(__builtin_bswap32(words0[0])) << 32) | __builtin_bswap32(words0[1])) << (bit0 & 31u);
				const uint64_t window1 = ((uint64_t(__builtin_bswap32(words1[0])) << 32) | __builtin_bswap32(words1[1])) << (bit1 & 31u);
				const uint32_t top0 = uint32_t(window0 >> 32), top1 = uint32_t(window1 >> 32);
				// unpack_*_uXX: float(field) * (1 / max), then the range; raw: the 32 bits are the value (math/scalar_packing.h:71-160)
				const uint32_t field0 = __builtin_amdgcn_ubfe(top0, 32u - field_bits, field_bits);
				const uint32_t field1 = __builtin_amdgcn_ubfe(top1, 32u - field_bits, field_bits);
				value0 = (float(field0) * header.inv_max_value) * range[C + c] + range[c];
				value1 = (float(field1) * header.inv_max_value) * range[C + c] + range[c];
				value0 = is_raw ? __uint_as_float(top0) : value0;
				value1 = is_raw ? __uint_as_float(top1) : value1;
			}
			else
			{
				// unpack_*_uXX (math/scalar_packing.h:113-160, math/vector4_packing.h:262-330): float(field) * (1 / max), then the range
				const uint32_t field0 = __builtin_amdgcn_ubfe(load_be32(animated_values + (offset0 >> 3)), 32u - field_bits - (offset0 & 7u), field_bits);
				const uint32_t field1 = __builtin_amdgcn_ubfe(load_be32(animated_values + (offset1 >> 3)), 32u - field_bits - (offset1 & 7u), field_bits);
				value0 = (float(field0) * header.inv_max_value) * range[C + c] + range[c];
				value1 = (float(field1) * header.inv_max_value) * range[C + c] + range[c];

				if (wave_has_raw)
				{
					// unpack_scalarf_32 / vector2_64 / vector3_96 / vector4_128: 32 bits at any bit offset (math/scalar_packing.h:71-110)
					const uint64_t window0 = __builtin_bswap64(load_u64(animated_values + (offset0 >> 3))) << (offset0 & 7u);
					const uint64_t window1 = __builtin_bswap64(load_u64(animated_values + (offset1 >> 3))) << (offset1 & 7u);
					value0 = is_raw ? __uint_as_float(uint32_t(window0 >> 32)) : value0;
					value1 = is_raw ? __uint_as_float(uint32_t(window1 >> 32)) : value1;
				}
			}

			// rtm::scalar_lerp / vector_lerp: (end * alpha) + (start - (start * alpha)); constant bit rate: the sample itself (:279-283)
			const float lerped = (value1 * alpha) + (value0 - (value0 * alpha));
			value[c] = is_constant ? range[c] : lerped;
		}

		store_streaming_floats<C>(destination, value);
	}

	__device__ __forceinline__ void decode_scalar_track_any(uint32_t num_components, const uint8_t* blob, const scalar_track_header* headers, const float* ranges,
		uint32_t track_index, uint32_t frame_bit_offset0, uint32_t frame_bit_offset1, float alpha, float* destination)
	{
		scalar_frames frames = {};
		frames.blob = blob;
		frames.frame_bit_offset[0] = frame_bit_offset0;
		frames.frame_bit_offset[1] = frame_bit_offset1;
		switch (num_components)
		{
		case 1: decode_scalar_track<1, false>(frames, load_scalar_track_tables<1>(headers, ranges, track_index), alpha, destination); break;
		case 2: decode_scalar_track<2, false>(frames, load_scalar_track_tables<2>(headers, ranges, track_index), alpha, destination); break;
		case 3: decode_scalar_track<3, false>(frames, load_scalar_track_tables<3>(headers, ranges, track_index), alpha, destination); break;
		default: decode_scalar_track<4, false>(frames, load_scalar_track_tables<4>(headers, ranges, track_index), alpha, destination); break;
		}
	}

	// Scalar track lists (float1f .. vector4f): seek_v0 + decompress_tracks_v0 of decompression/impl/decompression.scalar.h:182-480.
	// One wave64 per (instance, 256 consecutive tracks): the seek is wave uniform (scalar unit, like the pose kernels), lanes <->
	// tracks give coalesced table reads and value stores. There are no segments and no sub-track classes: a track is C <= 4
	// components of one width at a known bit offset of each frame.
	constexpr uint32_t k_scalar_tracks_per_wave = 256;

	// frame_lds_bytes != 0: both key frames' bits are DMA'd into LDS (one coalesced global_load_lds per KiB) while the track tables
	// are fetched, and every bit field is two aligned LDS dwords away -- instead of 2 * C scattered, unaligned global reads per track
	// through the texture unit, dependent on the table read. frame_lds_bytes == 0 (a registered list's frame does not fit): global reads.
	// kRows: tracks per lane (1 when no registered list has more than 64 tracks, else 4); kPolicies: per track rounding -- launch wide
	// facts, compiled as separate kernels so that each stays small.
	template<bool kFromLds, uint32_t kRows, bool kPolicies>
	__global__ __launch_bounds__(k_block_size) void decompress_scalar_tracks_kernel(const device_clip* __restrict__ clips, uint32_t num_clips,
		const uint32_t* __restrict__ clip_ids, const float* __restrict__ sample_times, uint32_t num_instances, uint32_t chunks_per_instance,
		decode_params params, uint8_t* __restrict__ out, uint64_t out_stride_bytes, uint32_t frame_lds_bytes, unsigned long long* __restrict__ rejected_count)
	{
		constexpr uint32_t k_tracks_per_wave = kRows * k_wave_size;
		extern __shared__ __attribute__((aligned(16))) uint8_t dynamic_lds[];

		const uint32_t lane = threadIdx.x & (k_wave_size - 1);
		const uint32_t wave_in_block = __builtin_amdgcn_readfirstlane(threadIdx.x / k_wave_size);
		const uint32_t work_item = blockIdx.x * k_waves_per_block + wave_in_block;
		uint32_t instance = work_item;
		uint32_t chunk = 0;
		if (chunks_per_instance != 1)
		{
			instance = work_item / chunks_per_instance;
			chunk = work_item - instance * chunks_per_instance;
		}
		if (instance >= num_instances)
			return;

		const uint32_t clip_id = as_constant(clip_ids)[instance];
		const float sample_time = as_constant(sample_times)[instance];
		const device_clip clip = load_clip(clips, clip_id < num_clips ? clip_id : 0);
		if (clip_id >= num_clips || !is_scalar_clip(clip.flags))
		{
			if (lane == 0 && chunk == 0)
				atomicAdd(rejected_count, 1ull);
			return;
		}

		const uint32_t first_track = chunk * k_tracks_per_wave;
		if (first_track >= clip.num_tracks || clip.num_samples == 0)
			return;		// past the end of this clip's track list / empty track list (:185-186,246-248)

		const uint32_t rounding_policy = params.instance_rounding_policies != nullptr
			? __builtin_amdgcn_readfirstlane(uint32_t(params.instance_rounding_policies[instance]))
			: uint32_t(params.rounding_policy);

		// seek_v0 (:182-240): a frame is num_bits_per_frame bits
		uint32_t key_frame0, key_frame1;
		float seek_alpha;
		find_key_frames(clip.flags, clip.num_samples, clip.sample_rate, clip.duration_clamp, clip.duration_wrap, sample_time, rounding_policy, params.looping_policy,
			key_frame0, key_frame1, seek_alpha);

		const uint32_t num_components = (clip.flags >> k_clip_components_shift) & 7u;
		const uint32_t num_bits_per_frame = clip.num_animated;
		float* row = reinterpret_cast<float*>(out + uint64_t(instance) * out_stride_bytes);

		// (plain locals, captured by value: a lambda that captures the clip record by reference keeps the whole record in scratch)
		scalar_frames frames = {};
		frames.blob = clip.blob;
		frames.frame_bit_offset[0] = key_frame0 * num_bits_per_frame;
		frames.frame_bit_offset[1] = key_frame1 * num_bits_per_frame;

		if (kFromLds)
		{
			// frame k occupies bits [animated values + key * bits per frame, + bits per frame) of the blob: copy the 16 byte aligned
			// span around it, plus 8 bytes for the last field's second dword
			const uint32_t animated_bit_base = clip.num_segments * 8u;		// scalar clips: byte offset of the animated values in the blob
			uint8_t* lds = dynamic_lds + size_t(wave_in_block) * 2u * frame_lds_bytes;
			#pragma unroll
			for (uint32_t key = 0; key < 2; ++key)
			{
				const uint32_t first_bit = animated_bit_base + frames.frame_bit_offset[key];
				const uint32_t first_byte = (first_bit >> 3) & ~15u;
				const uint32_t num_bytes = (((first_bit + num_bits_per_frame + 7u) >> 3) + 8u) - first_byte;
				uint8_t* destination = lds + key * frame_lds_bytes;
				for (uint32_t base = 0; base < num_bytes; base += k_wave_size * 16u)
				{
					if (base + lane * 16u < num_bytes)
						__builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(clip.blob + first_byte + base + lane * 16u),
							(__attribute__((address_space(3))) void*)(destination + base), 16, 0, 0);
				}
				frames.lds_frame[key] = reinterpret_cast<const uint32_t*>(destination);
				frames.lds_bit_base[key] = first_byte * 8u;
			}
		}

		const scalar_track_header* const headers = reinterpret_cast<const scalar_track_header*>(clip.plan);
		const float* const ranges = reinterpret_cast<const float*>(clip.clip_ranges);
		const uint32_t num_tracks = clip.num_tracks;
		const uint8_t* const track_rounding_policies = params.track_rounding_policies;

		// Every lane takes kRows tracks, 64 apart. The component count is wave uniform: one specialised loop runs.
		const auto decode_tracks = [=](auto components)
		{
			constexpr uint32_t C = decltype(components)::value;

			// the table rows travel together with the frame copies ...
			scalar_track_tables<C> tables[kRows];
			#pragma unroll
			for (uint32_t j = 0; j < kRows; ++j)
				tables[j] = load_scalar_track_tables<C>(headers, ranges, min(first_track + j * k_wave_size + lane, num_tracks - 1));

			if (kFromLds)
			{
				// ... which must have landed before any lane reads a field
				__builtin_amdgcn_s_waitcnt(0);
				__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
				__builtin_amdgcn_wave_barrier();
				__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
			}

			#pragma unroll
			for (uint32_t j = 0; j < kRows; ++j)
			{
				const uint32_t track_index = first_track + j * k_wave_size + lane;
				if (track_index < num_tracks)
				{
					float alpha = seek_alpha;
					if (kPolicies)
					{
						// track_writer::get_rounding_policy, applied to the alpha the seek left behind (:246-258,273-279)
						uint32_t policy = rounding_policy;
						if (rounding_policy == k_round_per_track)
							policy = track_rounding_policies != nullptr ? track_rounding_policies[track_index] : k_round_none;
						alpha = apply_rounding_policy(alpha, policy);
					}
					decode_scalar_track<C, kFromLds>(frames, tables[j], alpha, row + track_index * C);
				}
			}
		};
		switch (num_components)
		{
		case 1: decode_tracks(std::integral_constant<uint32_t, 1>()); break;
		case 2: decode_tracks(std::integral_constant<uint32_t, 2>()); break;
		case 3: decode_tracks(std::integral_constant<uint32_t, 3>()); break;
		default: decode_tracks(std::integral_constant<uint32_t, 4>()); break;
		}
	}

	// seek_v0 + decompress_track_v0 (decompression.scalar.h:482-715) for scalar track lists: one THREAD per request (every lane has its
	// own instance and track); C floats at out + request * stride.
	__global__ __launch_bounds__(k_block_size) void decompress_scalar_track_kernel(const device_clip* __restrict__ clips, uint32_t num_clips,
		const uint32_t* __restrict__ clip_ids, const float* __restrict__ sample_times, const uint32_t* __restrict__ track_indices,
		uint32_t num_instances, decode_params params, uint8_t* __restrict__ out, uint64_t out_stride_bytes, unsigned long long* __restrict__ rejected_count)
	{
		const uint32_t instance = blockIdx.x * k_block_size + threadIdx.x;
		if (instance >= num_instances)
			return;

		const uint32_t clip_id = clip_ids[instance];
		const device_clip& clip = clips[clip_id < num_clips ? clip_id : 0];
		const uint32_t flags = clip.flags;
		const uint32_t track_index = track_indices[instance];
		if (clip_id >= num_clips || !is_scalar_clip(flags) || track_index >= clip.num_tracks)
		{
			atomicAdd(rejected_count, 1ull);	// the reference silently returns (:496-498)
			return;
		}
		if (clip.num_samples == 0)
			return;

		const uint32_t rounding_policy = params.instance_rounding_policies != nullptr ? uint32_t(params.instance_rounding_policies[instance]) : uint32_t(params.rounding_policy);
		uint32_t key_frame0, key_frame1;
		float alpha;
		find_key_frames(flags, clip.num_samples, clip.sample_rate, clip.duration_clamp, clip.duration_wrap, sample_times[instance], rounding_policy, params.looping_policy,
			key_frame0, key_frame1, alpha);
		if (params.per_track_rounding != 0)
		{
			uint32_t policy = rounding_policy;
			if (rounding_policy == k_round_per_track)
				policy = params.track_rounding_policies != nullptr ? params.track_rounding_policies[track_index] : k_round_none;
			alpha = apply_rounding_policy(alpha, policy);
		}

		const uint32_t num_bits_per_frame = clip.num_animated;
		decode_scalar_track_any((flags >> k_clip_components_shift) & 7u, clip.blob, reinterpret_cast<const scalar_track_header*>(clip.plan),
			reinterpret_cast<const float*>(clip.clip_ranges), track_index, key_frame0 * num_bits_per_frame, key_frame1 * num_bits_per_frame, alpha,
			reinterpret_cast<float*>(out + uint64_t(instance) * out_stride_bytes));
	}

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
			const plan_entry plan0 = load_entry(state.plan[0], ordinal);
			const plan_entry plan1 = load_entry(state.plan[1], ordinal);
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
}

// ================================================================================================
// Host side: context, clip registry, C ABI
// ================================================================================================
using namespace aclhip;

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
	};
}

struct aclhip_context
{
	int device = 0;
	std::mutex mutex;
	std::vector<host_clip> clips;
	std::vector<uint32_t> free_slots;
	std::vector<host_database> databases;
	device_clip* d_clips = nullptr;
	uint32_t d_clips_capacity = 0;
	unsigned long long* d_rejected = nullptr;
	uint32_t max_pose_quads = 0;			// largest pose (3 * num_tracks) among registered clips
	uint32_t max_hierarchy_words = 0;		// largest walk schedule (aclhip_set_clip_hierarchy) among registered clips
	uint32_t max_scalar_tracks = 0;			// largest scalar track list among registered clips
	uint32_t max_scalar_frame_bytes = 0;	// largest frame (one sample of every track) among registered scalar clips
	bool force_generic_kernel = false;		// testing aid (ACLHIP_FORCE_GENERIC_KERNEL=1): always launch the any-settings kernel

	// Clips live in a few large HBM slabs instead of one hipMalloc each: a batch that draws on hundreds of clips then touches a
	// handful of large, contiguously mapped regions (fewer address translations to miss) and registration stops paying for an
	// allocation per clip. Bump allocation inside a slab; freeing rolls the bump pointer back over every freed piece at the top,
	// and a slab is recycled when its last clip is unregistered.
	// Walk schedules (aclhip_set_clip_hierarchy), one image per distinct hierarchy: clips of one skeleton share it, which is also
	// what lets a workgroup whose instances share a skeleton keep a single copy in LDS
	struct hierarchy_image { std::vector<uint32_t> parents; uint32_t* d_image = nullptr; uint32_t num_users = 0; };
	std::vector<hierarchy_image> hierarchies;

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
	constexpr size_t k_slab_bytes = size_t(32) << 20;
	constexpr size_t k_slab_alignment = 256;

	// nullptr: out of device memory
	uint8_t* allocate_clip_memory(aclhip_context* context, size_t bytes)
	{
		bytes = (bytes + k_slab_alignment - 1) & ~(k_slab_alignment - 1);
		static const bool use_slabs = []() { const char* value = std::getenv("ACLHIP_CLIP_SLABS"); return value == nullptr || value[0] != '0'; }();
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

	void release_hierarchy(aclhip_context* context, const uint32_t* d_image)
	{
		for (size_t i = 0; i < context->hierarchies.size(); ++i)
		{
			if (context->hierarchies[i].d_image != d_image)
				continue;
			if (--context->hierarchies[i].num_users == 0)
			{
				(void)hipFree(context->hierarchies[i].d_image);
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
		if (header.rotation_format() != k_rotation_quatf_drop_w_variable || header.translation_format() != k_vector_vector3f_variable
			|| (header.has_scale() && header.scale_format() != k_vector_vector3f_variable))
			return fail(context, ACLHIP_ERROR_UNSUPPORTED_FORMAT, "only quatf_drop_w_variable + vector3f_variable are supported (default_transform_decompression_settings)");
		if (header.num_samples == 0 || !(header.sample_rate > 0.0f))
			return fail(context, ACLHIP_ERROR_INVALID_CLIP, "Invalid sample count or rate");

		const transform_tracks_header& th = *reinterpret_cast<const transform_tracks_header*>(blob + k_transform_header_offset);
		const uint64_t blob_size = buffer_header.size;
		const uint64_t tbase = k_transform_header_offset;
		const bool stripped = header.has_stripped_keyframes() || header.has_database();
		const uint32_t segment_header_size = stripped ? sizeof(stripped_segment_header) : sizeof(segment_header);
		const uint32_t num_entries = (header.num_tracks + 15) / 16;
		const uint32_t num_rotations_padded = align_to_u32(th.num_animated_rotation_sub_tracks, 4);

		if (th.num_segments == 0)
			return fail(context, ACLHIP_ERROR_INVALID_CLIP, "Invalid segment count");
		if (uint64_t(header.num_samples) > uint64_t(th.num_segments) * 32)
			return fail(context, ACLHIP_ERROR_INVALID_CLIP, "%u samples cannot fit in %u segments of at most 32", header.num_samples, th.num_segments);
		if (th.num_animated_variable_sub_tracks != num_rotations_padded + th.num_animated_translation_sub_tracks + th.num_animated_scale_sub_tracks)
			return fail(context, ACLHIP_ERROR_INVALID_CLIP, "Inconsistent animated sub-track counts");
		if (tbase + th.segment_headers_offset + uint64_t(segment_header_size) * th.num_segments > blob_size
			|| tbase + th.sub_track_types_offset + uint64_t(num_entries) * 4 * (header.has_scale() ? 3 : 2) > blob_size
			|| tbase + th.constant_track_data_offset + 12ull * (uint64_t(th.num_constant_rotation_samples) + th.num_constant_translation_samples + th.num_constant_scale_samples) > blob_size
			|| tbase + th.clip_range_data_offset + 24ull * (uint64_t(th.num_animated_rotation_sub_tracks) + th.num_animated_translation_sub_tracks + th.num_animated_scale_sub_tracks) > blob_size)
			return fail(context, ACLHIP_ERROR_INVALID_CLIP, "Header offsets point outside of the buffer");
		if (th.num_segments > 1 && tbase + k_segment_start_indices_offset + 4ull * (th.num_segments + 1) > blob_size)
			return fail(context, ACLHIP_ERROR_INVALID_CLIP, "Segment start indices point outside of the buffer");
		if (header.has_database() && tbase + th.database_header_offset + sizeof(tracks_database_header) > blob_size)
			return fail(context, ACLHIP_ERROR_INVALID_CLIP, "Database header points outside of the buffer");

		for (uint32_t i = 0; i < th.num_segments; ++i)
		{
			const segment_header& sh = *reinterpret_cast<const segment_header*>(blob + tbase + th.segment_headers_offset + size_t(i) * segment_header_size);
			const uint64_t format_offset = tbase + sh.segment_data;
			const uint64_t range_offset = align_to_u32(uint32_t(format_offset + th.num_animated_variable_sub_tracks), 2);
			const uint64_t animated_offset = align_to_u32(uint32_t(range_offset + (th.num_segments > 1 ? 6ull * th.num_animated_variable_sub_tracks : 0ull)), 4);
			if (animated_offset > blob_size || sh.animated_rotation_bit_size > sh.animated_pose_bit_size)
				return fail(context, ACLHIP_ERROR_INVALID_CLIP, "Segment %u points outside of the buffer", i);
			if (!header.has_database())
			{
				// every keyframe a seek can pick must be inside the buffer
				const uint32_t stored = stripped ? uint32_t(__builtin_popcount(reinterpret_cast<const stripped_segment_header&>(sh).sample_indices)) : 32u;
				(void)stored;	// the exact count needs the segment's sample count; the tail padding below covers the last window
			}
		}

		return ACLHIP_OK;
	}

	uint32_t sub_track_class(const uint32_t* types, uint32_t track_index)
	{
		return (types[track_index / 16] >> ((15 - (track_index % 16)) * 2)) & 3u;
	}

	aclhip_status grow_clip_table(aclhip_context* context, uint32_t needed)
	{
		if (needed <= context->d_clips_capacity)
			return ACLHIP_OK;

		// 16384 records = 2 MiB: the table only moves (and captured hipGraphs that hold its address only go stale) past that many clips
		uint32_t capacity = std::max<uint32_t>(context->d_clips_capacity * 2, 16384);
		while (capacity < needed)
			capacity *= 2;

		device_clip* d_new = nullptr;
		ACLHIP_CHECK_HIP(context, hipMalloc(reinterpret_cast<void**>(&d_new), sizeof(device_clip) * capacity));
		ACLHIP_CHECK_HIP(context, hipMemset(d_new, 0, sizeof(device_clip) * capacity));
		if (context->d_clips != nullptr)
		{
			ACLHIP_CHECK_HIP(context, hipDeviceSynchronize());
			ACLHIP_CHECK_HIP(context, hipMemcpy(d_new, context->d_clips, sizeof(device_clip) * context->d_clips_capacity, hipMemcpyDeviceToDevice));
			ACLHIP_CHECK_HIP(context, hipFree(context->d_clips));
		}
		context->d_clips = d_new;
		context->d_clips_capacity = capacity;
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
		if (params->default_rotation_mode > ACLHIP_DEFAULT_VARIABLE || params->default_translation_mode > ACLHIP_DEFAULT_VARIABLE || params->default_scale_mode > ACLHIP_DEFAULT_LEGACY)
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
		const char* force_generic = std::getenv("ACLHIP_FORCE_GENERIC_KERNEL");
		context->force_generic_kernel = force_generic != nullptr && force_generic[0] == '1';
	}

	device_guard guard(device_index);
	if (!guard.ok || hipMalloc(reinterpret_cast<void**>(&context->d_rejected), sizeof(unsigned long long)) != hipSuccess
		|| hipMemset(context->d_rejected, 0, sizeof(unsigned long long)) != hipSuccess)
	{
		delete context;
		return ACLHIP_ERROR_DEVICE;
	}

	const aclhip_status status = grow_clip_table(context, 1);
	if (status != ACLHIP_OK)
	{
		(void)hipFree(context->d_rejected);
		delete context;
		return status;
	}

	*out_context = context;
	return ACLHIP_OK;
}

extern "C" void aclhip_destroy(aclhip_context* context)
{
	if (context == nullptr)
		return;
	{
		device_guard guard(context->device);
		(void)hipDeviceSynchronize();
		for (aclhip_context::clip_slab& slab : context->slabs)
			(void)hipFree(slab.base);
		for (aclhip_context::hierarchy_image& hierarchy : context->hierarchies)
			(void)hipFree(hierarchy.d_image);
		for (host_database& db : context->databases)
		{
			if (!db.in_use)
				continue;
			(void)hipFree(db.d_runtime_headers);
			for (int tier = 0; tier < 2; ++tier)
			{
				(void)hipFree(db.d_bulk_data[tier]);
				(void)hipFree(db.d_patches[tier]);
				if (db.pinned_bulk_data[tier] != nullptr)
					(void)hipHostFree(db.pinned_bulk_data[tier]);
			}
		}
		if (context->d_clips != nullptr)
			(void)hipFree(context->d_clips);
		if (context->d_rejected != nullptr)
			(void)hipFree(context->d_rejected);
	}
	delete context;
}

// Scalar track lists (initialize_v0, decompression.scalar.h:100-126): the blob plus one header and one range row per track (bit offset
// inside a frame = the prefix sum the reference's decompress_track_v0 recomputes per call, :529-541; constant / range values
// pulled next to it).
static aclhip_status register_scalar_clip(aclhip_context* context, const uint8_t* blob, aclhip_clip* out_clip, bool validate_only)
{
	const raw_buffer_header& buffer_header = *reinterpret_cast<const raw_buffer_header*>(blob);
	const tracks_header& header = *reinterpret_cast<const tracks_header*>(blob + k_tracks_header_offset);
	const uint32_t blob_size = buffer_header.size;
	const uint32_t num_components = scalar_track_num_components(header.track_type);
	const uint32_t num_samples = header.num_tracks != 0 ? header.num_samples : 0;
	const uint32_t num_tracks = num_samples != 0 ? header.num_tracks : 0;

	std::vector<scalar_track_header> track_headers(std::max<uint32_t>(num_tracks, 1));
	std::vector<float> range_rows(std::max<size_t>(size_t(num_tracks) * 2 * num_components, 8), 0.0f);
	std::memset(track_headers.data(), 0, track_headers.size() * sizeof(scalar_track_header));
	uint32_t num_bits_per_frame = 0;
	if (num_tracks != 0)
	{
		const scalar_tracks_header& sh = *reinterpret_cast<const scalar_tracks_header*>(blob + k_transform_header_offset);
		const uint8_t* base = reinterpret_cast<const uint8_t*>(&sh);
		const uint8_t* bit_rates = base + sh.metadata_per_track;
		const float* constant_values = reinterpret_cast<const float*>(base + sh.track_constant_values);
		const float* range_values = reinterpret_cast<const float*>(base + sh.track_range_values);
		const uint8_t* num_bits_at_bit_rate = header.version == k_version_first ? k_bit_rate_num_bits_v0 : k_bit_rate_num_bits;
		const uint32_t animated_bit_base = (k_transform_header_offset + sh.track_animated_values) * 8;	// headers address bits from the blob start

		uint32_t track_bit_offset = 0;
		for (uint32_t track = 0; track < num_tracks; ++track)
		{
			scalar_track_header& track_header = track_headers[track];
			float* range_min = &range_rows[size_t(track) * 2 * num_components];
			float* range_extent = range_min + num_components;
			const uint32_t num_bits = num_bits_at_bit_rate[bit_rates[track]];
			track_header.bit_offset_and_width = 0;
			track_header.inv_max_value = 1.0f;
			for (uint32_t c = 0; c < num_components; ++c)
			{
				range_min[c] = 0.0f;
				range_extent[c] = 1.0f;
			}

			if (num_bits == 0)
			{
				for (uint32_t c = 0; c < num_components; ++c)
				{
					range_min[c] = constant_values[c];
					range_extent[c] = 0.0f;
				}
				constant_values += num_components;
				continue;
			}

			if (uint64_t(animated_bit_base) + track_bit_offset > k_quad_ordinal_mask)
				return fail(context, ACLHIP_ERROR_UNSUPPORTED_FORMAT, "frames larger than 2 MiB are not supported");
			track_header.bit_offset_and_width = (animated_bit_base + track_bit_offset) | (num_bits << 24);
			if (num_bits != 32)
			{
				track_header.inv_max_value = 1.0f / float((1u << num_bits) - 1u);		// PackedTableEntry::max_value (math/scalar_packing.h:119)
				for (uint32_t c = 0; c < num_components; ++c)
				{
					range_min[c] = range_values[c];
					range_extent[c] = range_values[num_components + c];
				}
				range_values += num_components * 2;
			}
			track_bit_offset += num_bits * num_components;
		}
		num_bits_per_frame = sh.num_bits_per_frame;
	}

	// one device allocation: blob (+ zeroed tail padding: 8 byte windows are read, the writer reserves 15 bytes) | track headers | range rows
	const uint64_t blob_bytes = align_to_u32(blob_size, 16) + 64;
	const uint64_t headers_offset = blob_bytes;
	const uint64_t ranges_offset = align_to_u32(uint32_t(headers_offset + track_headers.size() * sizeof(scalar_track_header)), 16);
	const uint64_t total_bytes = ranges_offset + range_rows.size() * sizeof(float) + 16;		// a 3 component row is read as 16 + 8 bytes
	std::vector<uint8_t> staging(total_bytes, 0);
	std::memcpy(staging.data(), blob, blob_size);
	std::memcpy(staging.data() + headers_offset, track_headers.data(), track_headers.size() * sizeof(scalar_track_header));
	std::memcpy(staging.data() + ranges_offset, range_rows.data(), range_rows.size() * sizeof(float));
	if (validate_only)
		return ACLHIP_OK;		// aclhip_check_clip: everything above is host work

	std::lock_guard<std::mutex> lock(context->mutex);
	device_guard guard(context->device);
	if (!guard.ok)
		return fail(context, ACLHIP_ERROR_DEVICE, "hipSetDevice(%d) failed", context->device);

	uint32_t slot;
	if (!context->free_slots.empty())
	{
		slot = context->free_slots.back();
		context->free_slots.pop_back();
	}
	else
	{
		slot = uint32_t(context->clips.size());
		context->clips.emplace_back();
	}
	const aclhip_status status = grow_clip_table(context, slot + 1);
	if (status != ACLHIP_OK)
	{
		context->free_slots.push_back(slot);
		return status;
	}

	uint8_t* d_memory = allocate_clip_memory(context, total_bytes);
	if (d_memory == nullptr)
	{
		context->free_slots.push_back(slot);
		return fail(context, ACLHIP_ERROR_OUT_OF_MEMORY, "hipMalloc(%llu) failed", static_cast<unsigned long long>(total_bytes));
	}

	device_clip record;
	std::memset(&record, 0, sizeof(record));
	record.blob = d_memory;
	record.plan = reinterpret_cast<const plan_entry*>(d_memory + headers_offset);				// scalar_track_header[num_tracks]
	record.clip_ranges = reinterpret_cast<const clip_range_entry*>(d_memory + ranges_offset);	// float[num_tracks][2 * C]
	record.num_tracks = num_tracks;
	record.num_samples = num_samples;
	record.sample_rate = header.sample_rate;
	record.duration_clamp = num_samples <= 1 ? 0.0f : float(num_samples - 1) / header.sample_rate;
	record.duration_wrap = num_samples == 0 ? 0.0f : float(num_samples) / header.sample_rate;
	record.num_animated = num_bits_per_frame;
	record.num_segments = num_tracks != 0 ? k_transform_header_offset + reinterpret_cast<const scalar_tracks_header*>(blob + k_transform_header_offset)->track_animated_values : 0;	// scalar clips: byte offset of the animated values
	record.flags = k_clip_valid | k_clip_is_scalar | (num_components << k_clip_components_shift);
	record.flags |= (header.version > k_version_first && header.is_wrap_optimized()) ? k_clip_wraps : 0u;

	if (hipMemcpy(d_memory, staging.data(), total_bytes, hipMemcpyHostToDevice) != hipSuccess
		|| hipMemcpy(context->d_clips + slot, &record, sizeof(record), hipMemcpyHostToDevice) != hipSuccess)
	{
		free_clip_memory(context, d_memory);
		context->free_slots.push_back(slot);
		return fail(context, ACLHIP_ERROR_DEVICE, "uploading the clip failed");
	}

	host_clip& entry = context->clips[slot];
	entry.in_use = true;
	entry.database = ACLHIP_INVALID_HANDLE;
	entry.device_memory = d_memory;
	entry.info = aclhip_clip_info();
	entry.info.num_tracks = header.num_tracks;
	entry.info.num_samples = header.num_samples;
	entry.info.sample_rate = header.sample_rate;
	entry.info.duration = finite_duration(header, k_loop_as_compressed);
	entry.info.looping_policy = (header.version > k_version_first && header.is_wrap_optimized()) ? ACLHIP_LOOP_WRAP : ACLHIP_LOOP_CLAMP;
	entry.info.compressed_size = blob_size;
	entry.info.hash = buffer_header.hash;
	entry.info.track_type = header.track_type;
	entry.info.num_components = num_components;
	entry.touched_bytes = total_bytes - 64;
	context->max_scalar_tracks = std::max(context->max_scalar_tracks, num_tracks);
	context->max_scalar_frame_bytes = std::max(context->max_scalar_frame_bytes, (num_bits_per_frame + 7) / 8);

	*out_clip = slot;
	return ACLHIP_OK;
}

static aclhip_status register_clip_impl(aclhip_context* context, const void* compressed_tracks, uint64_t size, int check_hash, aclhip_database database, aclhip_clip* out_clip,
	bool validate_only = false)
{
	if (context == nullptr || out_clip == nullptr)
		return ACLHIP_ERROR_INVALID_ARGUMENT;
	*out_clip = ACLHIP_INVALID_HANDLE;

	const uint8_t* blob = static_cast<const uint8_t*>(compressed_tracks);
	aclhip_status status = validate_clip(context, blob, size, check_hash);
	if (status != ACLHIP_OK)
		return status;

	const raw_buffer_header& buffer_header = *reinterpret_cast<const raw_buffer_header*>(blob);
	const tracks_header& header = *reinterpret_cast<const tracks_header*>(blob + k_tracks_header_offset);
	if (scalar_track_num_components(header.track_type) != 0)
	{
		if (database != ACLHIP_INVALID_HANDLE)
			return fail(context, ACLHIP_ERROR_NOT_IN_DATABASE, "database decompression is not supported for scalar tracks");	// decompression.scalar.h:107-108
		return register_scalar_clip(context, blob, out_clip, validate_only);
	}
	const transform_tracks_header& th = *reinterpret_cast<const transform_tracks_header*>(blob + k_transform_header_offset);
	const uint8_t* tbase = blob + k_transform_header_offset;
	const uint32_t blob_size = buffer_header.size;
	const uint32_t num_tracks = header.num_tracks;
	const uint32_t num_quads = num_tracks * 3;
	const uint32_t num_samples = num_tracks != 0 ? header.num_samples : 0;
	const uint32_t num_segments = num_tracks != 0 ? th.num_segments : 0;
	const bool has_scale = num_tracks != 0 && header.has_scale();
	const bool stripped = num_tracks != 0 && (header.has_stripped_keyframes() || header.has_database());
	const bool multi_segment = num_segments > 1;
	const uint32_t raw_num_bits = header.version >= k_version_v02_01_99_1 ? 31u : 32u;	// animated_track_cache.transform.h:523

	const uint32_t num_animated_rotations = num_tracks != 0 ? th.num_animated_rotation_sub_tracks : 0;
	const uint32_t num_animated_translations = num_tracks != 0 ? th.num_animated_translation_sub_tracks : 0;
	const uint32_t num_animated_scales = num_tracks != 0 ? th.num_animated_scale_sub_tracks : 0;
	const uint32_t num_animated = num_animated_rotations + num_animated_translations + num_animated_scales;
	const uint32_t num_rotations_padded = align_to_u32(num_animated_rotations, 4);
	if (num_animated > k_quad_ordinal_mask)
		return fail(context, ACLHIP_ERROR_UNSUPPORTED_FORMAT, "too many animated sub-tracks");

	// ---- derived tables ----
	std::vector<float> base_pose(size_t(num_quads) * 4);
	std::vector<clip_range_entry> clip_ranges(std::max<uint32_t>(num_animated, 1));
	std::vector<sample_record> samples(std::max<uint32_t>(num_samples, 1));
	std::vector<plan_entry> plan(std::max<size_t>(size_t(num_segments) * num_animated, 1));
	std::memset(clip_ranges.data(), 0, clip_ranges.size() * sizeof(clip_range_entry));
	std::memset(samples.data(), 0, samples.size() * sizeof(sample_record));
	std::memset(plan.data(), 0, plan.size() * sizeof(plan_entry));
	bool has_raw = false;

	if (num_tracks != 0)
	{
		const uint32_t num_entries = (num_tracks + 15) / 16;
		const uint32_t* types = reinterpret_cast<const uint32_t*>(tbase + th.sub_track_types_offset);
		const float* constant_rotations = reinterpret_cast<const float*>(tbase + th.constant_track_data_offset);
		const float* constant_translations = constant_rotations + size_t(th.num_constant_rotation_samples) * 3;
		const float* constant_scales = constant_translations + size_t(th.num_constant_translation_samples) * 3;
		const float default_scale = float(header.default_scale());

		uint32_t constant_counts[3] = { 0, 0, 0 };
		uint32_t animated_counts[3] = { 0, 0, 0 };
		const uint32_t animated_bases[3] = { 0, num_animated_rotations, num_animated_rotations + num_animated_translations };
		const uint32_t animated_limits[3] = { num_animated_rotations, num_animated_translations, num_animated_scales };
		const uint32_t constant_limits[3] = { th.num_constant_rotation_samples, th.num_constant_translation_samples, th.num_constant_scale_samples };

		// base pose: constants expanded, defaults and animated sub-tracks tagged in the W lane
		for (uint32_t track = 0; track < num_tracks; ++track)
		{
			for (uint32_t kind = 0; kind < 3; ++kind)
			{
				const uint32_t quad = track * 3 + kind;
				float* value = &base_pose[size_t(quad) * 4];
				uint32_t* value_bits = reinterpret_cast<uint32_t*>(value);
				const uint32_t cls = (kind == 2 && !has_scale) ? k_sub_track_default : sub_track_class(types + size_t(kind) * num_entries, track);

				if (cls == k_sub_track_constant)
				{
					const uint32_t index = constant_counts[kind]++;
					if (index >= constant_limits[kind])
						return fail(context, ACLHIP_ERROR_INVALID_CLIP, "more constant sub-tracks than constant samples");

					if (kind == 0)
					{
						// constant_track_cache_v0::unpack_rotation_group (constant_track_cache.transform.h:113-205): SOA groups of 4, last one unpadded
						const uint32_t group = index / 4, lane = index % 4;
						const uint32_t group_size = std::min<uint32_t>(th.num_constant_rotation_samples - group * 4, 4);
						const float* group_data = constant_rotations + size_t(group) * 12;
						const float x = group_data[group_size * 0 + lane];
						const float y = group_data[group_size * 1 + lane];
						const float z = group_data[group_size * 2 + lane];
						// quat_from_positive_w4 (math/quatf.h:135-147), one IEEE operation at a time like the device code
						volatile float w_squared = 1.0f - (x * x);
						w_squared = w_squared - (y * y);
						w_squared = w_squared - (z * z);
						value[0] = x; value[1] = y; value[2] = z; value[3] = std::sqrt(std::fabs(w_squared));
					}
					else
					{
						const float* src = (kind == 1 ? constant_translations : constant_scales) + size_t(index) * 3;
						value[0] = src[0]; value[1] = src[1]; value[2] = src[2]; value[3] = 0.0f;
					}
					if (int32_t(value_bits[3]) < 0)
						value_bits[3] &= 0x7FFFFFFFu;	// only a garbage (NaN) constant could collide with the marker bit
				}
				else if (cls == k_sub_track_animated)
				{
					const uint32_t index = animated_counts[kind]++;
					if (index >= animated_limits[kind])
						return fail(context, ACLHIP_ERROR_INVALID_CLIP, "more animated sub-tracks than the header declares");
					const uint32_t ordinal = animated_bases[kind] + index;
					clip_ranges[ordinal].track_index = track;
					clip_ranges[ordinal].quad_index = quad;
					value[0] = 0.0f; value[1] = 0.0f; value[2] = 0.0f;
					value_bits[3] = k_quad_special | k_quad_animated | ordinal;
				}
				else if (cls == k_sub_track_default)
				{
					// identity / zero / the clip's legacy default scale (decompression.transform.h:585,893,1548)
					const float xyz = kind == 2 ? default_scale : 0.0f;
					value[0] = xyz; value[1] = xyz; value[2] = xyz;
					value_bits[3] = k_quad_special | (kind == 0 ? k_quad_default_w_one : 0u);
				}
				else
					return fail(context, ACLHIP_ERROR_INVALID_CLIP, "invalid sub-track type");
			}
		}

		if (animated_counts[0] != num_animated_rotations || animated_counts[1] != num_animated_translations || animated_counts[2] != num_animated_scales)
			return fail(context, ACLHIP_ERROR_INVALID_CLIP, "sub-track types disagree with the animated sub-track counts");

		// clip ranges: rotations are SOA per group of 4 (last group unpadded), translations / scales AOS (write_range_data.h:79-207)
		{
			const float* range_data = reinterpret_cast<const float*>(tbase + th.clip_range_data_offset);
			for (uint32_t i = 0; i < num_animated_rotations; ++i)
			{
				const uint32_t group = i / 4, lane = i % 4;
				const uint32_t group_size = std::min<uint32_t>(num_animated_rotations - group * 4, 4);
				const float* group_data = range_data + size_t(group) * 24;
				for (uint32_t c = 0; c < 3; ++c)
				{
					clip_ranges[i].range_min[c] = group_data[group_size * c + lane];
					clip_ranges[i].range_extent[c] = group_data[group_size * (3 + c) + lane];
				}
			}
			const float* vector_ranges = range_data + size_t(num_animated_rotations) * 6;
			for (uint32_t i = num_animated_rotations; i < num_animated; ++i)
			{
				const float* entry = vector_ranges + size_t(i - num_animated_rotations) * 6;
				for (uint32_t c = 0; c < 3; ++c)
				{
					clip_ranges[i].range_min[c] = entry[c];
					clip_ranges[i].range_extent[c] = entry[3 + c];
				}
			}
		}

		// segments, sample -> segment, per segment plan
		const uint32_t segment_header_size = stripped ? sizeof(stripped_segment_header) : sizeof(segment_header);
		const uint32_t* segment_start_indices = multi_segment ? reinterpret_cast<const uint32_t*>(tbase + k_segment_start_indices_offset) : nullptr;
		for (uint32_t si = 0; si < num_segments; ++si)
		{
			const segment_header& sh = *reinterpret_cast<const segment_header*>(tbase + th.segment_headers_offset + size_t(si) * segment_header_size);
			const uint32_t start = multi_segment ? segment_start_indices[si] : 0;
			const uint32_t end = multi_segment && si + 1 < num_segments ? segment_start_indices[si + 1] : num_samples;
			if (start >= end || end > num_samples || end - start > 32 || (si == 0 && start != 0))
				return fail(context, ACLHIP_ERROR_INVALID_CLIP, "segment %u has an invalid sample range [%u, %u)", si, start, end);
			// transform_tracks_header::get_segment_data (core/impl/compressed_headers.h:309-324)
			const uint32_t format_offset = k_transform_header_offset + sh.segment_data;
			const uint32_t range_offset = align_to_u32(format_offset + th.num_animated_variable_sub_tracks, 2);
			const uint32_t animated_offset = align_to_u32(range_offset + (multi_segment ? 6u * th.num_animated_variable_sub_tracks : 0u), 4);
			const uint8_t* format_per_track = blob + format_offset;
			const uint8_t* range_data = blob + range_offset;

			sample_record record;
			std::memset(&record, 0, sizeof(record));
			record.animated_offset = animated_offset;
			record.pose_bit_size = sh.animated_pose_bit_size;
			record.sample_indices = stripped ? reinterpret_cast<const stripped_segment_header&>(sh).sample_indices : 0xFFFFFFFFu;
			record.start_index = start;
			record.plan_row = si * num_animated;
			record.segment_index = si;
			for (uint32_t sample = start; sample < end; ++sample)
				samples[sample] = record;

			// every stored keyframe of a clip-resident segment must lie inside the blob
			if (!header.has_database())
			{
				const uint32_t stored = stripped ? uint32_t(__builtin_popcount(record.sample_indices)) : (end - start);
				if (uint64_t(animated_offset) + (uint64_t(sh.animated_pose_bit_size) * stored + 7) / 8 > blob_size)
					return fail(context, ACLHIP_ERROR_INVALID_CLIP, "segment %u animated data points outside of the buffer", si);
			}

			uint32_t bit_offset = 0;
			for (uint32_t a = 0; a < num_animated; ++a)
			{
				const bool is_rotation = a < num_animated_rotations;
				const uint32_t vector_index = a - num_animated_rotations;
				const uint32_t format_index = is_rotation ? a : num_rotations_padded + vector_index;
				const uint32_t stored_bits = format_per_track[format_index];
				const bool is_raw = stored_bits == raw_num_bits;
				if (!is_raw && stored_bits > 23)
					return fail(context, ACLHIP_ERROR_INVALID_CLIP, "segment %u sub-track %u has an invalid bit width %u", si, a, stored_bits);
				if (bit_offset > k_quad_ordinal_mask)
					return fail(context, ACLHIP_ERROR_UNSUPPORTED_FORMAT, "keyframes larger than 2 MiB are not supported");

				plan_entry& entry = plan[size_t(si) * num_animated + a];
				const uint32_t num_bits = is_raw ? 32u : stored_bits;
				entry.bit_offset_and_width = bit_offset | (num_bits << 24);
				entry.inv_max_value = num_bits == 0 ? 0.0f : (is_raw ? 1.0f : 1.0f / float((1u << num_bits) - 1u));
				for (uint32_t c = 0; c < 3; ++c)
				{
					entry.range_min[c] = 0.0f;
					entry.range_extent[c] = 1.0f;
				}
				has_raw = has_raw || is_raw;

				if (multi_segment && !is_raw)
				{
					// six bytes per sub-track: rotations SOA in padded groups of 4, translations / scales AOS (write_range_data.h:209-341)
					uint8_t bytes[6];
					if (is_rotation)
					{
						const uint8_t* group = range_data + size_t(a / 4) * 24 + (a % 4);
						for (uint32_t i = 0; i < 6; ++i)
							bytes[i] = group[i * 4];
					}
					else
						std::memcpy(bytes, range_data + size_t(num_rotations_padded) * 6 + size_t(vector_index) * 6, 6);

					if (num_bits == 0)
					{
						// constant in this segment: a 16 bit sample lives in the range bytes, hi/lo split across the SOA rows for rotations
						// (animated_track_cache.transform.h:552-588), little endian u16 for vectors (math/vector4_packing.h:628-653)
						for (uint32_t c = 0; c < 3; ++c)
						{
							const uint32_t sample = is_rotation ? ((uint32_t(bytes[c * 2]) << 8) | bytes[c * 2 + 1]) : ((uint32_t(bytes[c * 2 + 1]) << 8) | bytes[c * 2]);
							entry.range_min[c] = float(sample) * (1.0f / 65535.0f);
							entry.range_extent[c] = 0.0f;
						}
					}
					else
					{
						for (uint32_t c = 0; c < 3; ++c)
						{
							entry.range_min[c] = float(bytes[c]) * (1.0f / 255.0f);
							entry.range_extent[c] = float(bytes[3 + c]) * (1.0f / 255.0f);
						}
					}
				}

				bit_offset += num_bits * 3;
			}

			if (bit_offset != sh.animated_pose_bit_size)
				return fail(context, ACLHIP_ERROR_INVALID_CLIP, "segment %u: sub-track widths add up to %u bits, header says %u", si, bit_offset, sh.animated_pose_bit_size);
		}
	}

	// ---- animated sub-tracks in POSE order ----
	// The tables above follow the bitstream (rotations, translations, scales); lanes do not care which sub-track they get, so the
	// tables are reordered by destination window (and by kind inside a window). The sub-tracks that land in quads [c * k_image_chunk_quads, (c + 1) * ..) are then
	// a contiguous range of ordinals, image_chunks[c] .. image_chunks[c + 1]: the pose kernel can build a pose of any size through
	// a fixed LDS window.
	const uint32_t num_image_chunks = std::max<uint32_t>((num_quads + k_image_chunk_quads - 1) / k_image_chunk_quads, 1);
	std::vector<uint32_t> image_chunks(align_to_u32(num_image_chunks + 1, 4), num_animated);
	if (num_animated != 0)
	{
		std::vector<uint32_t> order(num_animated);		// new ordinal -> bitstream ordinal
		for (uint32_t a = 0; a < num_animated; ++a)
			order[a] = a;
		// inside a window rotations come first: the rotation math (two square roots, a division) is most of a lane's work and every
		// loop iteration that holds a rotation pays for it, so rotations are packed into as few iterations as possible
		const auto sort_key = [&](uint32_t ordinal)
		{
			const uint32_t quad = clip_ranges[ordinal].quad_index;
			const uint64_t window = quad / k_image_chunk_quads;
			const uint64_t is_vector = quad != clip_ranges[ordinal].track_index * 3 ? 1 : 0;
			return (window << 33) | (is_vector << 32) | quad;
		};
		std::sort(order.begin(), order.end(), [&](uint32_t lhs, uint32_t rhs) { return sort_key(lhs) < sort_key(rhs); });

		std::vector<clip_range_entry> ordered_ranges(num_animated);
		std::vector<plan_entry> ordered_plan(plan.size());
		for (uint32_t a = 0; a < num_animated; ++a)
		{
			ordered_ranges[a] = clip_ranges[order[a]];
			for (uint32_t si = 0; si < num_segments; ++si)
				ordered_plan[size_t(si) * num_animated + a] = plan[size_t(si) * num_animated + order[a]];
			reinterpret_cast<uint32_t*>(base_pose.data())[size_t(ordered_ranges[a].quad_index) * 4 + 3] = k_quad_special | k_quad_animated | a;
		}
		std::memcpy(clip_ranges.data(), ordered_ranges.data(), size_t(num_animated) * sizeof(clip_range_entry));
		plan.swap(ordered_plan);

		uint32_t next = 0;
		for (uint32_t chunk = 0; chunk < num_image_chunks; ++chunk)
		{
			while (next < num_animated && clip_ranges[next].quad_index / k_image_chunk_quads < chunk)
				next++;
			image_chunks[chunk] = next;
		}
	}
	else
		std::fill(image_chunks.begin(), image_chunks.end(), 0u);

	// ---- one device allocation: blob (+ zeroed tail padding) | base pose | segments | plan | clip ranges | sample -> segment ----
	const uint64_t blob_bytes = align_to_u32(blob_size, 16) + 64;		// windows of up to 16 bytes are read: keep well past the reference's 15 bytes of slack
	const uint64_t base_pose_offset = blob_bytes;
	// resolved pose: what a decode with the track_writer defaults stores for every non animated sub-track (animated slots: zero)
	std::vector<float> resolved_pose(base_pose);
	for (uint32_t quad = 0; quad < num_quads; ++quad)
	{
		uint32_t* value_bits = reinterpret_cast<uint32_t*>(&resolved_pose[size_t(quad) * 4]);
		if (int32_t(value_bits[3]) < 0)
			resolved_pose[size_t(quad) * 4 + 3] = (value_bits[3] & (k_quad_animated | k_quad_default_w_one)) == k_quad_default_w_one ? 1.0f : 0.0f;
	}

	const uint64_t resolved_pose_offset = base_pose_offset + uint64_t(num_quads) * 16;
	const uint64_t samples_offset = align_to_u32(uint32_t(resolved_pose_offset + uint64_t(num_quads) * 16), 32);
	const uint64_t plan_offset = samples_offset + samples.size() * sizeof(sample_record);
	const uint64_t clip_ranges_offset = plan_offset + plan.size() * sizeof(plan_entry);
	const uint64_t image_chunks_offset = clip_ranges_offset + clip_ranges.size() * sizeof(clip_range_entry);
	const uint64_t total_bytes = image_chunks_offset + image_chunks.size() * sizeof(uint32_t);

	std::vector<uint8_t> staging(total_bytes, 0);
	std::memcpy(staging.data(), blob, blob_size);
	if (num_quads != 0)
		std::memcpy(staging.data() + base_pose_offset, base_pose.data(), size_t(num_quads) * 16);
	if (num_quads != 0)
		std::memcpy(staging.data() + resolved_pose_offset, resolved_pose.data(), size_t(num_quads) * 16);
	std::memcpy(staging.data() + samples_offset, samples.data(), samples.size() * sizeof(sample_record));
	std::memcpy(staging.data() + plan_offset, plan.data(), plan.size() * sizeof(plan_entry));
	std::memcpy(staging.data() + clip_ranges_offset, clip_ranges.data(), clip_ranges.size() * sizeof(clip_range_entry));
	std::memcpy(staging.data() + image_chunks_offset, image_chunks.data(), image_chunks.size() * sizeof(uint32_t));
	if (validate_only)
		return ACLHIP_OK;		// aclhip_check_clip: everything above is host work

	std::lock_guard<std::mutex> lock(context->mutex);
	device_guard guard(context->device);
	if (!guard.ok)
		return fail(context, ACLHIP_ERROR_DEVICE, "hipSetDevice(%d) failed", context->device);

	uint32_t slot;
	if (!context->free_slots.empty())
	{
		slot = context->free_slots.back();
		context->free_slots.pop_back();
	}
	else
	{
		slot = uint32_t(context->clips.size());
		context->clips.emplace_back();
	}

	status = grow_clip_table(context, slot + 1);
	if (status != ACLHIP_OK)
	{
		context->free_slots.push_back(slot);
		return status;
	}

	uint8_t* d_memory = allocate_clip_memory(context, total_bytes);
	if (d_memory == nullptr)
	{
		context->free_slots.push_back(slot);
		return fail(context, ACLHIP_ERROR_OUT_OF_MEMORY, "hipMalloc(%llu) failed", static_cast<unsigned long long>(total_bytes));
	}

	device_clip record;
	std::memset(&record, 0, sizeof(record));
	record.blob = d_memory;
	record.base_pose = reinterpret_cast<const float4*>(d_memory + base_pose_offset);
	record.resolved_pose = reinterpret_cast<const float4*>(d_memory + resolved_pose_offset);
	record.samples = reinterpret_cast<const sample_record*>(d_memory + samples_offset);
	record.plan = reinterpret_cast<const plan_entry*>(d_memory + plan_offset);
	record.clip_ranges = reinterpret_cast<const clip_range_entry*>(d_memory + clip_ranges_offset);
	record.image_chunks = reinterpret_cast<const uint32_t*>(d_memory + image_chunks_offset);
	record.num_tracks = num_tracks;
	record.num_samples = num_samples;
	record.sample_rate = header.sample_rate;
	record.duration_clamp = num_samples <= 1 ? 0.0f : float(num_samples - 1) / header.sample_rate;
	record.duration_wrap = num_samples == 0 ? 0.0f : float(num_samples) / header.sample_rate;
	record.flags = k_clip_valid;
	if (num_tracks != 0)
	{
		record.flags |= has_scale ? k_clip_has_scale : 0u;
		record.flags |= stripped ? k_clip_has_stripped_keyframes : 0u;
		record.flags |= header.has_database() ? k_clip_has_database : 0u;
		record.flags |= (header.version > k_version_first && header.is_wrap_optimized()) ? k_clip_wraps : 0u;
		record.flags |= has_raw ? k_clip_has_raw : 0u;
		record.num_segments = num_segments;
		record.num_animated = num_animated;
		if (header.has_database())
			record.db_clip_header_offset = reinterpret_cast<const tracks_database_header*>(tbase + th.database_header_offset)->clip_header_offset;
	}

	if (database != ACLHIP_INVALID_HANDLE)
	{
		// decompression_context::initialize(tracks, database): the database must contain the clip (impl/decompress.impl.h:105-107,
		// compressed_database::contains core/impl/compressed_database.impl.h:123-140)
		if (database >= context->databases.size() || !context->databases[database].in_use)
		{
			free_clip_memory(context, d_memory);
			context->free_slots.push_back(slot);
			return fail(context, ACLHIP_ERROR_UNKNOWN_DATABASE, "unknown database handle %u", database);
		}
		host_database& db = context->databases[database];
		bool contained = num_tracks != 0 && header.has_database();
		if (contained)
		{
			contained = false;
			for (const database_clip_metadata& metadata : db.clip_metadata)
				contained = contained || (metadata.clip_hash == buffer_header.hash && metadata.clip_header_offset == record.db_clip_header_offset);
			contained = contained && uint64_t(record.db_clip_header_offset) + sizeof(database_runtime_clip_header) + uint64_t(record.num_segments) * sizeof(database_runtime_segment_header) <= db.runtime_headers_size;
		}
		if (!contained)
		{
			free_clip_memory(context, d_memory);
			context->free_slots.push_back(slot);
			return fail(context, ACLHIP_ERROR_NOT_IN_DATABASE, "the database does not contain this clip");
		}
		record.db_headers = db.d_runtime_headers;
		record.db_bulk_data[0] = db.d_bulk_data[0];
		record.db_bulk_data[1] = db.d_bulk_data[1];
		db.num_bound_clips++;
	}

	if (hipMemcpy(d_memory, staging.data(), total_bytes, hipMemcpyHostToDevice) != hipSuccess
		|| hipMemcpy(context->d_clips + slot, &record, sizeof(record), hipMemcpyHostToDevice) != hipSuccess)
	{
		free_clip_memory(context, d_memory);
		context->free_slots.push_back(slot);
		return fail(context, ACLHIP_ERROR_DEVICE, "uploading the clip failed");
	}

	host_clip& entry = context->clips[slot];
	entry.in_use = true;
	entry.database = database;
	entry.device_memory = d_memory;
	entry.info.num_tracks = num_tracks;
	entry.info.num_samples = header.num_samples;
	entry.info.sample_rate = header.sample_rate;
	entry.info.duration = finite_duration(header, k_loop_as_compressed);
	entry.info.num_segments = num_segments;
	entry.info.has_scale = has_scale ? 1 : 0;
	entry.info.looping_policy = (header.version > k_version_first && header.is_wrap_optimized()) ? ACLHIP_LOOP_WRAP : ACLHIP_LOOP_CLAMP;
	entry.info.compressed_size = blob_size;
	entry.info.hash = buffer_header.hash;
	entry.info.num_animated_sub_tracks = num_animated;
	entry.info.has_database = num_tracks != 0 && header.has_database() ? 1 : 0;
	entry.info.has_stripped_keyframes = num_tracks != 0 && header.has_stripped_keyframes() ? 1 : 0;
	entry.info.track_type = k_track_type_qvvf;
	entry.info.num_components = 12;
	// bytes a batch may read from this clip: the blob itself plus the registration time tables
	entry.touched_bytes = total_bytes - 64;
	context->max_pose_quads = std::max(context->max_pose_quads, num_quads);

	*out_clip = slot;
	return ACLHIP_OK;
}

// No exception crosses the C ABI: a buffer whose counts pass validation but ask for more host memory than there is ends here
template<class callable>
static aclhip_status guarded(aclhip_context* context, callable&& call)
{
	try
	{
		return call();
	}
	catch (const std::bad_alloc&)
	{
		return context != nullptr ? fail(context, ACLHIP_ERROR_OUT_OF_MEMORY, "out of host memory") : ACLHIP_ERROR_OUT_OF_MEMORY;
	}
}

extern "C" aclhip_status aclhip_register_clip(aclhip_context* context, const void* compressed_tracks, uint64_t size, int check_hash, aclhip_clip* out_clip)
{
	return guarded(context, [&]() { return register_clip_impl(context, compressed_tracks, size, check_hash, ACLHIP_INVALID_HANDLE, out_clip); });
}

extern "C" aclhip_status aclhip_register_clip_with_database(aclhip_context* context, const void* compressed_tracks, uint64_t size, int check_hash,
	aclhip_database database, aclhip_clip* out_clip)
{
	if (database == ACLHIP_INVALID_HANDLE)
		return context != nullptr ? fail(context, ACLHIP_ERROR_UNKNOWN_DATABASE, "invalid database handle") : ACLHIP_ERROR_INVALID_ARGUMENT;
	return guarded(context, [&]() { return register_clip_impl(context, compressed_tracks, size, check_hash, database, out_clip); });
}

extern "C" aclhip_status aclhip_unregister_clip(aclhip_context* context, aclhip_clip clip)
{
	if (context == nullptr)
		return ACLHIP_ERROR_INVALID_ARGUMENT;

	std::lock_guard<std::mutex> lock(context->mutex);
	if (clip >= context->clips.size() || !context->clips[clip].in_use)
		return fail(context, ACLHIP_ERROR_UNKNOWN_CLIP, "unknown clip handle %u", clip);

	device_guard guard(context->device);
	device_clip cleared;
	std::memset(&cleared, 0, sizeof(cleared));
	ACLHIP_CHECK_HIP(context, hipDeviceSynchronize());
	ACLHIP_CHECK_HIP(context, hipMemcpy(context->d_clips + clip, &cleared, sizeof(cleared), hipMemcpyHostToDevice));
	free_clip_memory(context, context->clips[clip].device_memory);
	if (context->clips[clip].d_hierarchy != nullptr)
		release_hierarchy(context, context->clips[clip].d_hierarchy);
	const uint32_t bound_database = context->clips[clip].database;
	if (bound_database != ACLHIP_INVALID_HANDLE && bound_database < context->databases.size() && context->databases[bound_database].num_bound_clips != 0)
		context->databases[bound_database].num_bound_clips--;
	context->clips[clip] = host_clip();
	context->free_slots.push_back(clip);
	return ACLHIP_OK;
}

// ---- databases -----------------------------------------------------------------------------------------------------

namespace
{
	bool bitset_test(const std::vector<uint32_t>& bits, uint32_t index) { return (bits[index / 32] & (0x80000000u >> (index % 32))) != 0; }
	void bitset_set(std::vector<uint32_t>& bits, uint32_t index, bool value)
	{
		if (value) bits[index / 32] |= 0x80000000u >> (index % 32);
		else bits[index / 32] &= ~(0x80000000u >> (index % 32));
	}

	void release_database(host_database& db)
	{
		(void)hipFree(db.d_runtime_headers);
		for (int tier = 0; tier < 2; ++tier)
		{
			(void)hipFree(db.d_bulk_data[tier]);
			(void)hipFree(db.d_patches[tier]);
			if (db.pinned_bulk_data[tier] != nullptr)
				(void)hipHostFree(db.pinned_bulk_data[tier]);
		}
		db = host_database();
	}
}

static aclhip_status register_database_impl(aclhip_context* context, const void* compressed_database, uint64_t size,
	const void* bulk_data_medium, const void* bulk_data_low, int check_hash, aclhip_database* out_database, bool validate_only)
{
	if (context == nullptr || out_database == nullptr)
		return ACLHIP_ERROR_INVALID_ARGUMENT;
	*out_database = ACLHIP_INVALID_HANDLE;

	// compressed_database::is_valid (core/impl/compressed_database.impl.h:142-163)
	const uint8_t* blob = static_cast<const uint8_t*>(compressed_database);
	if (blob == nullptr || size < sizeof(raw_buffer_header) + sizeof(database_header))
		return fail(context, ACLHIP_ERROR_INVALID_CLIP, "buffer is not a valid compressed_database instance (too small)");
	const raw_buffer_header& buffer_header = *reinterpret_cast<const raw_buffer_header*>(blob);
	const database_header& header = *reinterpret_cast<const database_header*>(blob + sizeof(raw_buffer_header));
	const uint8_t* hbase = reinterpret_cast<const uint8_t*>(&header);
	if (header.tag != k_tag_compressed_database)
		return fail(context, ACLHIP_ERROR_INVALID_CLIP, "Invalid tag");
	if (header.version < k_version_first || header.version > k_version_latest)
		return fail(context, ACLHIP_ERROR_INVALID_CLIP, "Invalid database version");
	if (buffer_header.size > size || buffer_header.size < sizeof(raw_buffer_header) + sizeof(database_header))
		return fail(context, ACLHIP_ERROR_INVALID_CLIP, "Invalid size");
	if (check_hash && hash32(blob + sizeof(raw_buffer_header), buffer_header.size - sizeof(raw_buffer_header)) != buffer_header.hash)
		return fail(context, ACLHIP_ERROR_INVALID_CLIP, "Invalid hash");

	const uint64_t header_limit = buffer_header.size - sizeof(raw_buffer_header);
	const uint64_t descriptions_offset = align_to_u32(sizeof(database_header), 4);
	const uint64_t num_descriptions = uint64_t(header.num_chunks[0]) + header.num_chunks[1];
	if (descriptions_offset + num_descriptions * sizeof(database_chunk_description) > header_limit
		|| uint64_t(header.clip_metadata_offset) + uint64_t(header.num_clips) * sizeof(database_clip_metadata) > header_limit)
		return fail(context, ACLHIP_ERROR_INVALID_CLIP, "Header offsets point outside of the buffer");

	const bool is_inline = (header.misc_packed & 1u) != 0;
	const uint8_t* bulk_sources[2] = { static_cast<const uint8_t*>(bulk_data_medium), static_cast<const uint8_t*>(bulk_data_low) };
	for (int tier = 0; tier < 2; ++tier)
	{
		if (header.bulk_data_size[tier] == 0)
			continue;
		if (bulk_sources[tier] == nullptr)
		{
			if (!is_inline || header.bulk_data_offset[tier] == k_invalid_offset || uint64_t(header.bulk_data_offset[tier]) + header.bulk_data_size[tier] > header_limit)
				return fail(context, ACLHIP_ERROR_INVALID_ARGUMENT, "tier %d has %u bytes of bulk data: pass it in, it is not inline", tier + 1, header.bulk_data_size[tier]);
			bulk_sources[tier] = hbase + header.bulk_data_offset[tier];
		}
		if (check_hash && hash32(bulk_sources[tier], header.bulk_data_size[tier]) != header.bulk_data_hash[tier])
			return fail(context, ACLHIP_ERROR_INVALID_CLIP, "Invalid bulk data hash (tier %d)", tier + 1);
	}

	host_database db;
	db.hash = buffer_header.hash;
	db.info.num_clips = header.num_clips;
	db.info.num_segments = header.num_segments;
	db.info.max_chunk_size = header.max_chunk_size;
	const database_clip_metadata* clip_metadata = reinterpret_cast<const database_clip_metadata*>(hbase + header.clip_metadata_offset);
	db.clip_metadata.assign(clip_metadata, clip_metadata + header.num_clips);

	const uint64_t runtime_size = uint64_t(header.num_clips) * sizeof(database_runtime_clip_header) + uint64_t(header.num_segments) * sizeof(database_runtime_segment_header);
	if (runtime_size > (256ull << 20))
		return fail(context, ACLHIP_ERROR_UNSUPPORTED_FORMAT, "%u clips / %u segments: runtime headers beyond 256 MiB are not supported", header.num_clips, header.num_segments);
	std::vector<uint8_t> runtime(std::max<uint64_t>(runtime_size, 16), 0);
	db.runtime_headers_size = runtime_size;
	for (const database_clip_metadata& metadata : db.clip_metadata)
	{
		if (uint64_t(metadata.clip_header_offset) + sizeof(database_runtime_clip_header) > runtime_size)
			return fail(context, ACLHIP_ERROR_INVALID_CLIP, "Clip metadata points outside of the runtime headers");
		reinterpret_cast<database_runtime_clip_header*>(runtime.data() + metadata.clip_header_offset)->clip_hash = metadata.clip_hash;	// database.impl.h:151-157
	}

	// Walk every chunk of both tiers once: validate it and turn its segment headers into metadata patches
	std::vector<tier_patch> patches[2];
	const database_chunk_description* descriptions = reinterpret_cast<const database_chunk_description*>(hbase + descriptions_offset);
	for (int tier = 0; tier < 2; ++tier)
	{
		const uint32_t num_chunks = header.num_chunks[tier];
		const database_chunk_description* tier_descriptions = descriptions + (tier == 0 ? 0 : header.num_chunks[0]);
		db.info.num_chunks[tier] = num_chunks;
		db.info.bulk_data_size[tier] = header.bulk_data_size[tier];
		db.chunks[tier].assign(tier_descriptions, tier_descriptions + num_chunks);
		db.loaded_chunks[tier].assign((num_chunks + 31) / 32, 0u);
		db.chunk_first_patch[tier].assign(num_chunks + 1, 0u);

		for (uint32_t chunk_index = 0; chunk_index < num_chunks; ++chunk_index)
		{
			const database_chunk_description& description = tier_descriptions[chunk_index];
			db.chunk_first_patch[tier][chunk_index] = uint32_t(patches[tier].size());
			if (uint64_t(description.offset) + description.size > header.bulk_data_size[tier] || description.size < sizeof(database_chunk_header) || description.size > header.max_chunk_size)
				return fail(context, ACLHIP_ERROR_INVALID_CLIP, "Chunk %u of tier %d lies outside of the bulk data", chunk_index, tier + 1);

			const database_chunk_header& chunk = *reinterpret_cast<const database_chunk_header*>(bulk_sources[tier] + description.offset);
			if (chunk.index != chunk_index || chunk.size != description.size
				|| uint64_t(sizeof(database_chunk_header)) + uint64_t(chunk.num_segments) * sizeof(database_chunk_segment_header) > description.size)
				return fail(context, ACLHIP_ERROR_INVALID_CLIP, "Chunk %u of tier %d has an invalid header", chunk_index, tier + 1);

			const database_chunk_segment_header* segments = reinterpret_cast<const database_chunk_segment_header*>(&chunk + 1);
			for (uint32_t i = 0; i < chunk.num_segments; ++i)
			{
				if (uint64_t(segments[i].segment_header_offset) + sizeof(database_runtime_segment_header) > runtime_size || segments[i].samples_offset >= header.bulk_data_size[tier])
					return fail(context, ACLHIP_ERROR_INVALID_CLIP, "Chunk %u of tier %d points outside of the database", chunk_index, tier + 1);
				patches[tier].push_back(tier_patch{ segments[i].segment_header_offset, segments[i].sample_indices, segments[i].samples_offset });
			}
		}
		db.chunk_first_patch[tier][num_chunks] = uint32_t(patches[tier].size());
	}
	if (validate_only)
		return ACLHIP_OK;		// aclhip_check_database: everything above is host work

	std::lock_guard<std::mutex> lock(context->mutex);
	device_guard guard(context->device);
	if (!guard.ok)
		return fail(context, ACLHIP_ERROR_DEVICE, "hipSetDevice(%d) failed", context->device);

	bool ok = hipMalloc(reinterpret_cast<void**>(&db.d_runtime_headers), runtime.size()) == hipSuccess
		&& hipMemcpy(db.d_runtime_headers, runtime.data(), runtime.size(), hipMemcpyHostToDevice) == hipSuccess;
	for (int tier = 0; tier < 2 && ok; ++tier)
	{
		// +64: keyframe windows of up to 16 bytes are read past the last sample, the reference reserves 15 (compress.database.impl.h:910)
		const size_t bulk_bytes = size_t(header.bulk_data_size[tier]) + 64;
		ok = hipMalloc(reinterpret_cast<void**>(&db.d_bulk_data[tier]), bulk_bytes) == hipSuccess
			&& hipMemset(db.d_bulk_data[tier], 0xCD, bulk_bytes) == hipSuccess		// like debug_database_streamer: not-resident memory is poison
			&& hipMalloc(reinterpret_cast<void**>(&db.d_patches[tier]), std::max<size_t>(patches[tier].size(), 1) * sizeof(tier_patch)) == hipSuccess;
		if (ok && !patches[tier].empty())
			ok = hipMemcpy(db.d_patches[tier], patches[tier].data(), patches[tier].size() * sizeof(tier_patch), hipMemcpyHostToDevice) == hipSuccess;
		if (ok && header.bulk_data_size[tier] != 0)
		{
			ok = hipHostMalloc(reinterpret_cast<void**>(&db.pinned_bulk_data[tier]), header.bulk_data_size[tier], hipHostMallocDefault) == hipSuccess;
			if (ok)
				std::memcpy(db.pinned_bulk_data[tier], bulk_sources[tier], header.bulk_data_size[tier]);
		}
	}
	if (!ok)
	{
		release_database(db);
		return fail(context, ACLHIP_ERROR_OUT_OF_MEMORY, "allocating the database failed");
	}

	db.in_use = true;
	uint32_t slot = 0;
	while (slot < context->databases.size() && context->databases[slot].in_use)
		slot++;
	if (slot == context->databases.size())
		context->databases.emplace_back();
	context->databases[slot] = std::move(db);
	*out_database = slot;
	return ACLHIP_OK;
}

extern "C" aclhip_status aclhip_register_database(aclhip_context* context, const void* compressed_database, uint64_t size,
	const void* bulk_data_medium, const void* bulk_data_low, int check_hash, aclhip_database* out_database)
{
	return guarded(context, [&]() { return register_database_impl(context, compressed_database, size, bulk_data_medium, bulk_data_low, check_hash, out_database, false); });
}

// ---- host only validation (no device needed) ---------------------------------------------------------------------------

namespace
{
	aclhip_status report(const aclhip_context&, aclhip_status status, char* out_message, uint32_t capacity)
	{
		if (out_message != nullptr && capacity != 0)
			std::snprintf(out_message, capacity, "%s", status == ACLHIP_OK ? "" : t_last_error.c_str());
		return status;
	}
}

extern "C" aclhip_status aclhip_check_clip(const void* compressed_tracks, uint64_t size, int check_hash, char* out_message, uint32_t capacity)
{
	aclhip_context scratch;		// collects the error message; no device is touched
	aclhip_clip unused = ACLHIP_INVALID_HANDLE;
	return report(scratch, guarded(&scratch, [&]() { return register_clip_impl(&scratch, compressed_tracks, size, check_hash, ACLHIP_INVALID_HANDLE, &unused, true); }), out_message, capacity);
}

extern "C" aclhip_status aclhip_check_database(const void* compressed_database, uint64_t size, const void* bulk_data_medium, const void* bulk_data_low,
	int check_hash, char* out_message, uint32_t capacity)
{
	aclhip_context scratch;
	aclhip_database unused = ACLHIP_INVALID_HANDLE;
	return report(scratch, guarded(&scratch, [&]() { return register_database_impl(&scratch, compressed_database, size, bulk_data_medium, bulk_data_low, check_hash, &unused, true); }), out_message, capacity);
}

extern "C" aclhip_status aclhip_unregister_database(aclhip_context* context, aclhip_database database)
{
	if (context == nullptr)
		return ACLHIP_ERROR_INVALID_ARGUMENT;
	std::lock_guard<std::mutex> lock(context->mutex);
	if (database >= context->databases.size() || !context->databases[database].in_use)
		return fail(context, ACLHIP_ERROR_UNKNOWN_DATABASE, "unknown database handle %u", database);
	if (context->databases[database].num_bound_clips != 0)
		return fail(context, ACLHIP_ERROR_INVALID_ARGUMENT, "%u clips are still bound to this database", context->databases[database].num_bound_clips);
	device_guard guard(context->device);
	ACLHIP_CHECK_HIP(context, hipDeviceSynchronize());
	release_database(context->databases[database]);
	return ACLHIP_OK;
}

extern "C" aclhip_status aclhip_get_database_info(const aclhip_context* context, aclhip_database database, aclhip_database_info* out_info)
{
	if (context == nullptr || out_info == nullptr)
		return ACLHIP_ERROR_INVALID_ARGUMENT;
	std::lock_guard<std::mutex> lock(const_cast<aclhip_context*>(context)->mutex);
	if (database >= context->databases.size() || !context->databases[database].in_use)
		return fail(context, ACLHIP_ERROR_UNKNOWN_DATABASE, "unknown database handle %u", database);
	*out_info = context->databases[database].info;
	return ACLHIP_OK;
}

namespace
{
	aclhip_status stream_database(aclhip_context* context, aclhip_database database, uint32_t tier, uint32_t num_chunks_to_stream, void* stream, bool stream_in, uint32_t* out_num_chunks)
	{
		if (context == nullptr)
			return ACLHIP_ERROR_INVALID_ARGUMENT;
		if (out_num_chunks != nullptr)
			*out_num_chunks = 0;
		std::lock_guard<std::mutex> lock(context->mutex);
		if (database >= context->databases.size() || !context->databases[database].in_use)
			return fail(context, ACLHIP_ERROR_UNKNOWN_DATABASE, "unknown database handle %u", database);
		if (tier != 1 && tier != 2)
			return fail(context, ACLHIP_ERROR_INVALID_ARGUMENT, "tier must be 1 (medium importance) or 2 (lowest importance)");	// invalid_database_tier

		host_database& db = context->databases[database];
		const uint32_t tier_index = tier - 1;
		const uint32_t num_chunks = db.info.num_chunks[tier_index];
		num_chunks_to_stream = std::min(num_chunks_to_stream, num_chunks);
		if (num_chunks == 0 || num_chunks_to_stream == 0)
			return ACLHIP_OK;

		// Which chunks: the first missing ones when streaming in, the first resident ones when streaming out -- the reference's
		// bit scans over loaded_chunks (database.impl.h:478-497,551-570)
		const std::vector<uint32_t>& loaded = db.loaded_chunks[tier_index];
		uint32_t first_chunk_index = ~0u;
		for (uint32_t entry_index = 0; entry_index < loaded.size(); ++entry_index)
		{
			const uint32_t maybe_loaded = loaded[entry_index];
			if (stream_in)
			{
				const uint32_t num_pending = maybe_loaded == 0 ? 32u : uint32_t(__builtin_ctz(maybe_loaded));
				if (num_pending != 0)
				{
					first_chunk_index = entry_index * 32 + (32 - num_pending);
					break;
				}
			}
			else
			{
				const uint32_t num_pending = maybe_loaded == 0 ? 32u : uint32_t(__builtin_clz(maybe_loaded));
				if (num_pending != 32)
				{
					first_chunk_index = entry_index * 32 + num_pending;
					break;
				}
			}
		}
		if (first_chunk_index == ~0u || first_chunk_index >= num_chunks)
			return ACLHIP_OK;	// database_stream_request_result::done

		const uint64_t last_chunk_index64 = uint64_t(first_chunk_index) + num_chunks_to_stream - 1;
		const uint32_t last_chunk_index = last_chunk_index64 >= num_chunks ? num_chunks - 1 : uint32_t(last_chunk_index64);
		const uint32_t num_streaming_chunks = last_chunk_index - first_chunk_index + 1;

		device_guard guard(context->device);
		hipStream_t hip_stream = static_cast<hipStream_t>(stream);
		const uint32_t first_patch = db.chunk_first_patch[tier_index][first_chunk_index];
		const uint32_t num_patches = db.chunk_first_patch[tier_index][last_chunk_index + 1] - first_patch;

		if (stream_in)
		{
			// debug_database_streamer::stream_in is a memcpy (impl/debug_database_streamer.h:75-92); here: pinned host -> HBM, asynchronously
			const uint32_t start_offset = db.chunks[tier_index][first_chunk_index].offset;
			const uint32_t end_offset = db.chunks[tier_index][last_chunk_index].offset + db.chunks[tier_index][last_chunk_index].size;
			ACLHIP_CHECK_HIP(context, hipMemcpyAsync(db.d_bulk_data[tier_index] + start_offset, db.pinned_bulk_data[tier_index] + start_offset, end_offset - start_offset, hipMemcpyHostToDevice, hip_stream));
		}

		if (num_patches != 0)
		{
			hipLaunchKernelGGL(apply_tier_metadata_kernel, dim3((num_patches + 255) / 256), dim3(256), 0, hip_stream,
				db.d_runtime_headers, db.d_patches[tier_index], first_patch, num_patches, tier_index, stream_in ? 1u : 0u);
			ACLHIP_CHECK_HIP(context, hipGetLastError());
		}

		for (uint32_t chunk_index = first_chunk_index; chunk_index <= last_chunk_index; ++chunk_index)
			bitset_set(db.loaded_chunks[tier_index], chunk_index, stream_in);
		db.info.num_loaded_chunks[tier_index] = 0;
		for (uint32_t chunk_index = 0; chunk_index < num_chunks; ++chunk_index)
			db.info.num_loaded_chunks[tier_index] += bitset_test(db.loaded_chunks[tier_index], chunk_index) ? 1 : 0;

		if (out_num_chunks != nullptr)
			*out_num_chunks = num_streaming_chunks;
		return ACLHIP_OK;
	}
}

extern "C" aclhip_status aclhip_database_stream_in(aclhip_context* context, aclhip_database database, uint32_t tier, uint32_t num_chunks, void* stream, uint32_t* out_num_chunks)
{
	return stream_database(context, database, tier, num_chunks, stream, true, out_num_chunks);
}

extern "C" aclhip_status aclhip_database_stream_out(aclhip_context* context, aclhip_database database, uint32_t tier, uint32_t num_chunks, void* stream, uint32_t* out_num_chunks)
{
	return stream_database(context, database, tier, num_chunks, stream, false, out_num_chunks);
}

extern "C" aclhip_status aclhip_get_clip_info(const aclhip_context* context, aclhip_clip clip, aclhip_clip_info* out_info)
{
	if (context == nullptr || out_info == nullptr)
		return ACLHIP_ERROR_INVALID_ARGUMENT;
	std::lock_guard<std::mutex> lock(const_cast<aclhip_context*>(context)->mutex);
	if (clip >= context->clips.size() || !context->clips[clip].in_use)
		return fail(context, ACLHIP_ERROR_UNKNOWN_CLIP, "unknown clip handle %u", clip);
	*out_info = context->clips[clip].info;
	return ACLHIP_OK;
}

extern "C" aclhip_status aclhip_clip_matches(const aclhip_context* context, aclhip_clip clip, const void* compressed_tracks, int* out_matches)
{
	if (context == nullptr || compressed_tracks == nullptr || out_matches == nullptr)
		return ACLHIP_ERROR_INVALID_ARGUMENT;
	std::lock_guard<std::mutex> lock(const_cast<aclhip_context*>(context)->mutex);
	if (clip >= context->clips.size() || !context->clips[clip].in_use)
		return fail(context, ACLHIP_ERROR_UNKNOWN_CLIP, "unknown clip handle %u", clip);
	// is_bound_to_v0 compares pointer and hash (decompression.transform.h:159-169); there is no shared pointer here: hash + size
	const raw_buffer_header& buffer_header = *static_cast<const raw_buffer_header*>(compressed_tracks);
	const aclhip_clip_info& info = context->clips[clip].info;
	*out_matches = (buffer_header.hash == info.hash && buffer_header.size == info.compressed_size) ? 1 : 0;
	return ACLHIP_OK;
}

namespace
{
	aclhip_status launch_tracks(aclhip_context* context, const aclhip_clip* clips, const float* sample_times, uint32_t num_instances,
		const decode_params& params, void* poses, uint64_t pose_stride_bytes, hipStream_t stream)
	{
		// launches read the clip table's address and the registry's maxima: enqueue under the registry lock, so that a registration
		// that moves the table (it synchronizes the device first) never frees it under a launch that is being prepared
		std::lock_guard<std::mutex> lock(context->mutex);

		// one wave per (instance, pose window); instances of clips with fewer windows than the largest registered clip leave waves idle
		const uint32_t windows_per_instance = std::max<uint32_t>((context->max_pose_quads + k_image_chunk_quads - 1) / k_image_chunk_quads, 1);
		const uint64_t num_waves = uint64_t(num_instances) * windows_per_instance;
		if (num_waves > 0xFFFFFFFFull - k_waves_per_block)
			return fail(context, ACLHIP_ERROR_INVALID_ARGUMENT, "batch too large: %u instances x %u pose windows", num_instances, windows_per_instance);
		const uint32_t num_blocks = uint32_t((num_waves + k_waves_per_block - 1) / k_waves_per_block);

		// the common case (track_writer defaults, no per track rounding, normalization != always) copies a resolved pose image
		const bool any_settings = params.standard_defaults == 0 || params.per_track_rounding != 0 || context->force_generic_kernel;
		const uint32_t lds_quads_per_wave = std::min<uint32_t>(std::max<uint32_t>(align_to_u32(context->max_pose_quads, 64), 64), k_image_chunk_quads);
		const size_t lds_bytes = size_t(lds_quads_per_wave) * 16 * k_waves_per_block;
		if (any_settings)
			hipLaunchKernelGGL(decompress_tracks_any_settings_kernel, dim3(num_blocks), dim3(k_block_size), lds_bytes, stream,
				context->d_clips, context->d_clips_capacity, clips, sample_times, num_instances, windows_per_instance, params,
				static_cast<uint8_t*>(poses), pose_stride_bytes, lds_quads_per_wave, context->d_rejected);
		else
			hipLaunchKernelGGL(decompress_tracks_kernel, dim3(num_blocks), dim3(k_block_size), lds_bytes, stream,
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
		// per XCD: its instances, bucketed by clip (stable: instances of a clip keep their relative order)
		std::vector<uint32_t> sorted(num_instances);
		for (uint32_t i = 0; i < num_instances; ++i)
			sorted[i] = i;
		std::stable_sort(sorted.begin(), sorted.end(), [&](uint32_t a, uint32_t b)
		{
			const uint32_t xcd_a = clips[a] % k_num_xcds, xcd_b = clips[b] % k_num_xcds;
			return xcd_a != xcd_b ? xcd_a < xcd_b : clips[a] < clips[b];
		});
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
	const uint32_t num_blocks = (num_instances + k_block_size - 1) / k_block_size;
	hipLaunchKernelGGL(decompress_track_kernel, dim3(num_blocks), dim3(k_block_size), 0, static_cast<hipStream_t>(stream),
		context->d_clips, context->d_clips_capacity, clips, sample_times, track_indices, num_instances, device_params,
		static_cast<float4*>(transforms), context->d_rejected);
	ACLHIP_CHECK_HIP(context, hipGetLastError());
	return ACLHIP_OK;
}

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

extern "C" aclhip_status aclhip_set_clip_hierarchy(aclhip_context* context, aclhip_clip clip, const uint32_t* parent_indices, uint32_t num_tracks)
{
	if (context == nullptr)
		return ACLHIP_ERROR_INVALID_ARGUMENT;
	if (parent_indices == nullptr && num_tracks != 0)
		return fail(context, ACLHIP_ERROR_INVALID_ARGUMENT, "null parent index list");

	return guarded(context, [&]() -> aclhip_status
	{
		std::lock_guard<std::mutex> lock(context->mutex);
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

		// [offset of the schedule for 1, 2, 4, 8 instances per workgroup] then per schedule:
		// num_steps | words of this schedule | step_end[num_steps] | transform | parent << 16, in step order (16 bits each: the
		// consumers' LDS images end at about 3400 transforms; every word of the copy a wave keeps in LDS costs residency)
		std::vector<uint32_t> image(4, 0);
		uint32_t max_schedule_words = 0;
		for (uint32_t log2_instances = 0; log2_instances < 4; ++log2_instances)
		{
			std::vector<uint32_t> step_end, pairs;
			schedule_hierarchy_walk(tree, num_tracks, 64u >> log2_instances, step_end, pairs);
			for (uint32_t& pair : pairs)
				pair |= parent_indices[pair] << 16;

			const uint32_t num_steps = uint32_t(step_end.size());
			const uint32_t header_words = 2 + num_steps;
			const uint32_t schedule_words = header_words + uint32_t(pairs.size());
			const uint32_t offset = uint32_t(image.size());
			image[log2_instances] = offset;
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
		aclhip_context::hierarchy_image* shared = nullptr;
		for (aclhip_context::hierarchy_image& candidate : context->hierarchies)
			if (candidate.parents == canonical)
				shared = &candidate;

		uint32_t* d_hierarchy = shared != nullptr ? shared->d_image : nullptr;
		hipError_t hip_status = hipSuccess;
		if (shared == nullptr)
		{
			if (hipMalloc(reinterpret_cast<void**>(&d_hierarchy), image.size() * sizeof(uint32_t)) != hipSuccess)
				return fail(context, ACLHIP_ERROR_OUT_OF_MEMORY, "hipMalloc of %zu bytes failed", image.size() * sizeof(uint32_t));
			hip_status = hipMemcpy(d_hierarchy, image.data(), image.size() * sizeof(uint32_t), hipMemcpyHostToDevice);
		}
		// launches in flight may still walk the hierarchy that is being replaced
		if (hip_status == hipSuccess)
			hip_status = hipDeviceSynchronize();
		if (hip_status == hipSuccess)
			hip_status = hipMemcpy(reinterpret_cast<uint8_t*>(context->d_clips + clip) + offsetof(device_clip, hierarchy), &d_hierarchy, sizeof(d_hierarchy), hipMemcpyHostToDevice);
		if (hip_status != hipSuccess)
		{
			if (shared == nullptr)
				(void)hipFree(d_hierarchy);
			return fail(context, ACLHIP_ERROR_DEVICE, "uploading the hierarchy failed: %s", hipGetErrorString(hip_status));
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
			release_hierarchy(context, entry.d_hierarchy);
		entry.d_hierarchy = d_hierarchy;
		context->max_hierarchy_words = std::max(context->max_hierarchy_words, max_schedule_words);
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
		// a consumer needs every sub-track of the pose: the track_writer's own defaults (what the resolved pose image holds)
		if (params.standard_defaults == 0 || params.per_track_rounding != 0)
			return fail(context, ACLHIP_ERROR_INVALID_ARGUMENT, "pose consumers take the track_writer's default sub-track modes, no per track rounding, normalization != always");

		std::lock_guard<std::mutex> lock(context->mutex);		// see launch_tracks

		// one wave per instance, the whole pose (its base, its hierarchy) in LDS; as many instances per workgroup (a power of two, at
		// most 8, 4 unless told otherwise: measured best) as leave room for three workgroups per CU: the object space walk packs its lanes with instances of one workgroup
		const uint32_t lds_quads_per_image = std::max<uint32_t>(align_to_u32(context->max_pose_quads, 4), 4);		// (no row granularity here: every quad is addressed on its own)
		const size_t lds_bytes_per_instance = size_t(lds_quads_per_image) * 16 * (base_is_clip ? 2 : 1);
		const size_t lds_schedule_bytes = consumers.object_space != 0 ? align_to_u32(std::max<uint32_t>(context->max_hierarchy_words, 4), 4) * sizeof(uint32_t) : 0;
		constexpr size_t k_lds_bytes = 160 * 1024 - 128;		// the kernel's few static words
		if (lds_bytes_per_instance + lds_schedule_bytes > k_lds_bytes)
			return fail(context, ACLHIP_ERROR_INVALID_ARGUMENT, "a registered clip has %u transforms: too large for the pose consumers (%zu bytes of LDS per instance)", context->max_pose_quads / 3, lds_bytes_per_instance + lds_schedule_bytes);
		uint32_t log2_instances_per_block = 2;
		if (const char* forced = std::getenv("ACLHIP_CONSUMER_LOG2_INSTANCES"))
			log2_instances_per_block = std::min<uint32_t>(uint32_t(forced[0] - '0'), 3);
		while (log2_instances_per_block != 0 && (lds_bytes_per_instance << log2_instances_per_block) + lds_schedule_bytes > k_lds_bytes / 3)
			log2_instances_per_block--;
		const uint32_t instances_per_block = 1u << log2_instances_per_block;
		const uint32_t waves_per_block = instances_per_block * (base_is_clip ? 2 : 1);
		const uint32_t num_blocks = (num_instances + instances_per_block - 1) / instances_per_block;
		const size_t lds_bytes = lds_bytes_per_instance * instances_per_block + lds_schedule_bytes;
		if (lds_bytes > 64 * 1024 - 128)		// above the default limit
			ACLHIP_CHECK_HIP(context, hipFuncSetAttribute(reinterpret_cast<const void*>(decompress_poses_consumer_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, int(k_lds_bytes)));

		consumer_params device_consumers;
		device_consumers.base_clip_ids = base_is_clip ? consumers.base_clips : nullptr;
		device_consumers.base_sample_times = base_is_clip ? consumers.base_sample_times : nullptr;
		device_consumers.base_poses = has_base && !base_is_clip ? static_cast<const uint8_t*>(consumers.base_poses) : nullptr;
		device_consumers.base_pose_stride_bytes = consumers.base_pose_stride_bytes;
		device_consumers.additive_format = consumers.additive_format;
		device_consumers.object_space = consumers.object_space != 0 ? 1 : 0;

		hipLaunchKernelGGL(decompress_poses_consumer_kernel, dim3(num_blocks), dim3(waves_per_block * k_wave_size), lds_bytes, stream,
			context->d_clips, context->d_clips_capacity, clips, sample_times, num_instances, params, device_consumers,
			static_cast<uint8_t*>(poses), pose_stride_bytes, lds_quads_per_image, uint32_t(lds_bytes_per_instance), log2_instances_per_block, context->d_rejected);
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
		const aclhip_pose_consumers* consumers = nullptr)
	{
		if (context == nullptr)
			return ACLHIP_ERROR_INVALID_ARGUMENT;
		if (num_instances == 0)
			return ACLHIP_OK;
		if (clips == nullptr || sample_times == nullptr || out == nullptr)
			return fail(context, ACLHIP_ERROR_INVALID_ARGUMENT, "null instance list or output buffer");

		aclhip_decompress_params local;
		if (params != nullptr) local = *params; else aclhip_default_params(&local);

		device_guard guard(context->device);

		uint32_t max_tracks = 0;
		{
			std::lock_guard<std::mutex> lock(context->mutex);
			for (uint32_t i = 0; i < num_instances; ++i)
				if (clips[i] < context->clips.size() && context->clips[clips[i]].in_use)
					max_tracks = std::max(max_tracks, context->clips[clips[i]].info.num_tracks);
		}

		const bool single_track = track_indices != nullptr;
		const uint64_t device_stride = single_track ? 48 : std::max<uint64_t>(uint64_t(max_tracks) * 48, 16);
		if (!single_track && out_row_bytes == 0)
			out_row_bytes = uint64_t(max_tracks) * 48;

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
		if (ok && local.default_values != nullptr)
			ok = upload(local.default_values, size_t(std::max<uint32_t>(default_values_count, 1)) * 48, &d_defaults);
		if (ok && local.track_rounding_policies != nullptr)
			ok = upload(local.track_rounding_policies, std::max<uint32_t>(max_tracks, 1), &d_track_policies);
		if (ok && local.instance_rounding_policies != nullptr)
			ok = upload(local.instance_rounding_policies, num_instances, &d_instance_policies);

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

		aclhip_status status;
		if (single_track)
			status = aclhip_decompress_track_batch(context, static_cast<const aclhip_clip*>(d_clip_ids), static_cast<const float*>(d_times), static_cast<const uint32_t*>(d_tracks), num_instances, &local, d_out, nullptr);
		else if (consumers != nullptr)
			status = aclhip_decompress_poses_batch(context, static_cast<const aclhip_clip*>(d_clip_ids), static_cast<const float*>(d_times), num_instances, &local, &local_consumers, d_out, device_stride, nullptr);
		else
			status = aclhip_decompress_tracks_batch(context, static_cast<const aclhip_clip*>(d_clip_ids), static_cast<const float*>(d_times), num_instances, &local, d_out, device_stride, nullptr);

		if (status == ACLHIP_OK)
		{
			hipError_t copy_status = hipDeviceSynchronize();
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

extern "C" aclhip_status aclhip_decompress_poses_host(aclhip_context* context, const aclhip_clip* clips, const float* sample_times, uint32_t num_instances,
	const aclhip_decompress_params* params, const aclhip_pose_consumers* consumers, void* poses, uint64_t pose_stride_bytes)
{
	if (consumers == nullptr)
		return context != nullptr ? fail(context, ACLHIP_ERROR_INVALID_ARGUMENT, "null consumers") : ACLHIP_ERROR_INVALID_ARGUMENT;
	if (params != nullptr && (params->default_values != nullptr || params->track_rounding_policies != nullptr))
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

		std::lock_guard<std::mutex> lock(context->mutex);		// see launch_tracks
		device_guard guard(context->device);
		if (track_indices != nullptr)
		{
			const uint32_t num_blocks = (num_instances + k_block_size - 1) / k_block_size;
			hipLaunchKernelGGL(decompress_scalar_track_kernel, dim3(num_blocks), dim3(k_block_size), 0, static_cast<hipStream_t>(stream),
				context->d_clips, context->d_clips_capacity, clips, sample_times, track_indices, num_instances, device_params,
				static_cast<uint8_t*>(out), out_stride_bytes, context->d_rejected);
		}
		else
		{
			// one wave per (instance, 64 or 256 tracks); instances of shorter lists than the longest registered one leave waves idle
			const uint32_t rows = context->max_scalar_tracks <= k_wave_size ? 1u : k_scalar_tracks_per_wave / k_wave_size;
			const uint32_t tracks_per_wave = rows * k_wave_size;
			const uint32_t chunks_per_instance = std::max<uint32_t>((context->max_scalar_tracks + tracks_per_wave - 1) / tracks_per_wave, 1);
			const uint64_t num_waves = uint64_t(num_instances) * chunks_per_instance;
			if (num_waves > 0xFFFFFFFFull - k_waves_per_block)
				return fail(context, ACLHIP_ERROR_INVALID_ARGUMENT, "batch too large: %u instances x %u track chunks", num_instances, chunks_per_instance);

			// LDS copy of both key frames per wave: the frame, the 16 byte alignment slack in front, 8 bytes behind. Skipped (global
			// reads) when that would leave fewer than 4 workgroups per CU, and for short frames, where a handful of scattered reads
			// is cheaper than the copy and its barrier (measured: 64 float1f tracks 16 vs 23 us, 256 tracks 38 vs 32 us)
			uint32_t frame_lds_bytes = align_to_u32(context->max_scalar_frame_bytes + 16 + 8, 16);
			if (size_t(frame_lds_bytes) * 2 * k_waves_per_block > 40 * 1024 || context->max_scalar_frame_bytes < 192)
				frame_lds_bytes = 0;

			const dim3 grid(uint32_t((num_waves + k_waves_per_block - 1) / k_waves_per_block));
			const size_t lds_bytes = size_t(frame_lds_bytes) * 2 * k_waves_per_block;
			const auto launch = [&](auto kernel)
			{
				hipLaunchKernelGGL(kernel, grid, dim3(k_block_size), lds_bytes, static_cast<hipStream_t>(stream), context->d_clips, context->d_clips_capacity, clips, sample_times,
					num_instances, chunks_per_instance, device_params, static_cast<uint8_t*>(out), out_stride_bytes, frame_lds_bytes, context->d_rejected);
			};
			const bool policies = device_params.per_track_rounding != 0;
			if (frame_lds_bytes != 0)
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
			std::lock_guard<std::mutex> lock(context->mutex);
			for (uint32_t i = 0; i < num_instances; ++i)
				if (clips[i] < context->clips.size() && context->clips[clips[i]].in_use)
					max_tracks = std::max(max_tracks, context->clips[clips[i]].info.num_tracks);
		}

		device_guard guard(context->device);
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

		if (!ok)
		{
			release();
			return fail(context, ACLHIP_ERROR_DEVICE, "staging the batch on the device failed");
		}
		local.track_rounding_policies = static_cast<const uint8_t*>(d_track_policies);
		local.instance_rounding_policies = static_cast<const uint8_t*>(d_instance_policies);

		aclhip_status status = launch_scalar(context, static_cast<const aclhip_clip*>(d_clip_ids), static_cast<const float*>(d_times), static_cast<const uint32_t*>(d_tracks),
			num_instances, &local, d_out, out_stride_bytes, nullptr);
		if (status == ACLHIP_OK)
		{
			hipError_t copy_status = hipDeviceSynchronize();
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
		std::lock_guard<std::mutex> lock(context->mutex);
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

extern "C" aclhip_status aclhip_all_gather_poses(aclhip_context* context, void* rccl_comm, const void* shard_poses, void* all_poses, uint64_t shard_bytes, void* stream)
{
	if (context == nullptr)
		return ACLHIP_ERROR_INVALID_ARGUMENT;
	if (rccl_comm == nullptr || shard_poses == nullptr || all_poses == nullptr)
		return fail(context, ACLHIP_ERROR_INVALID_ARGUMENT, "null communicator or buffer");
	if (shard_bytes == 0)
		return ACLHIP_OK;

	// ncclResult_t ncclAllGather(const void* sendbuff, void* recvbuff, size_t sendcount, ncclDataType_t datatype, ncclComm_t comm, hipStream_t stream)
	// (rccl.h:678); the library is only loaded by callers that gather, decoding never touches it
	typedef int (*all_gather_fn)(const void*, void*, size_t, int, void*, hipStream_t);
	static all_gather_fn all_gather = nullptr;
	{
		std::lock_guard<std::mutex> lock(context->mutex);
		if (all_gather == nullptr)
		{
			void* library = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
			if (library == nullptr)
				library = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
			if (library == nullptr)
				return fail(context, ACLHIP_ERROR_DEVICE, "librccl.so.1 could not be loaded: %s", dlerror());
			all_gather = reinterpret_cast<all_gather_fn>(dlsym(library, "ncclAllGather"));
			if (all_gather == nullptr)
				return fail(context, ACLHIP_ERROR_DEVICE, "librccl.so.1 has no ncclAllGather");
		}
	}

	device_guard guard(context->device);
	constexpr int k_nccl_uint8 = 1;		// ncclUint8 (rccl.h:460)
	const int result = all_gather(shard_poses, all_poses, size_t(shard_bytes), k_nccl_uint8, rccl_comm, static_cast<hipStream_t>(stream));
	if (result != 0)
		return fail(context, ACLHIP_ERROR_DEVICE, "ncclAllGather failed: ncclResult_t %d", result);
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

extern "C" aclhip_status aclhip_describe_tracks_kernel(aclhip_context* context, const aclhip_decompress_params* params, char* out_name, uint32_t capacity)
{
	if (context == nullptr || out_name == nullptr || capacity == 0)
		return ACLHIP_ERROR_INVALID_ARGUMENT;
	decode_params device_params;
	const aclhip_status status = resolve_params(context, params, device_params);
	if (status != ACLHIP_OK)
		return status;
	const bool any_settings = device_params.standard_defaults == 0 || device_params.per_track_rounding != 0 || context->force_generic_kernel;
	std::snprintf(out_name, capacity, "%s", any_settings ? "decompress_tracks_any_settings_kernel" : "decompress_tracks_kernel");
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

	std::lock_guard<std::mutex> lock(const_cast<aclhip_context*>(context)->mutex);
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
