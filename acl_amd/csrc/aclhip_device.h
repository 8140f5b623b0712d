// aclhip_device.h -- gfx950 device functions shared by the decode kernels: per-instance seek and the decode of
// one animated sub-track from the ACL bitstream. No host code here.
//
// Arithmetic follows the reference operation by operation (fp32, round to nearest, NO contraction -- this file
// must be compiled with -ffp-contract=off) so that poses match the reference CPU decoder bit for bit:
//   seek_v0                         decompression/impl/decompression.transform.h:206-563
//   unpack_animated_quat            decompression/impl/animated_track_cache.transform.h:515-687
//   unpack_animated_vector3         decompression/impl/animated_track_cache.transform.h:871-990
//   remap_segment/clip_range_data4  decompression/impl/animated_track_cache.transform.h:302-350,391-466
//   quat_from_positive_w4 / quat_lerp_no_normalization4 / quat_normalize4   math/quatf.h:135-211
//   unpack_vector3_uXX / _96 / _u48 / _u24                                   math/vector4_packing.h:479-599,628-653,781-818,921-1035
// (paths relative to /root/reference/includes/acl)
//
// What differs from the reference is WHERE the per clip bookkeeping happens: everything that only depends on
// (clip, segment) -- which segment a sample lives in, where each animated sub-track's bits start inside a keyframe,
// its width, its segment range as floats -- is worked out once at registration and kept next to the blob in HBM
// (sample_record / plan_entry / clip_range_entry below). The CPU walks the format bytes serially per pose
// (count_animated_group_bit_size); a wave here looks the answers up.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "acl_format.h"

// Clip tables are read-only for the lifetime of a launch: reading them through the constant address space lets wave uniform
// accesses go through the scalar cache (s_load) and per lane accesses through global_load instead of flat_load.
#define ACLHIP_CONSTANT __attribute__((address_space(4)))

namespace aclhip
{
	template<class T>
	__device__ __forceinline__ const ACLHIP_CONSTANT T* as_constant(const T* pointer) { return (const ACLHIP_CONSTANT T*)pointer; }

	// One per SAMPLE (a copy of its segment's facts), 16 bytes, read through the scalar cache: one load tells a wave everything seek
	// needs about the keyframe it lands on. (32 bytes per sample in round 1: at 16 the records of a few hundred clips stay resident
	// in an XCD's L2 next to everything else a mixed batch keeps re-reading.)
	struct alignas(16) sample_record
	{
		uint32_t animated_offset;		// first stored keyframe of the sample's segment, from the blob start (4 byte aligned)
		uint32_t pose_bit_size;			// bits per stored keyframe (segment_header::animated_pose_bit_size)
		uint32_t sample_indices;		// clips with stripped keyframes / a database: stored keyframes of the segment, MSB = first sample (a segment then
										// holds at most 32 samples). Every other clip: the index of the sample inside its segment, whole -- a clip in
										// the full formats is ONE segment of any length (compress.transform.impl.h:168-176: no segmenting without a variable format)
		uint32_t segment_and_local;		// segment index << 5 | (index of the sample inside its segment) & 31
	};

	// Clips bound to a compressed_database keep 32 bytes per sample: the record above and a COPY of the runtime tier metadata of the
	// sample's segment -- (samples_offset << 32) | sample_indices for the medium and the lowest importance tier, 0 while a tier is not
	// resident (database_runtime_segment_header, core/impl/compressed_headers.h:420-439). The seek then needs what it needs for any clip:
	// one scalar load per key (round 2 read the runtime header itself, whose address is only known once the sample record has arrived:
	// a fourth dependent load, 3.0 instead of 1.8 us of seek). refresh_database_sample_tiers_kernel rewrites the copies behind every
	// stream_in / stream_out, stream ordered; each word is written and read whole, so a decode racing on another stream sees the old or
	// the new state of a tier like the reference's relaxed atomics do (database.impl.h:616-618).
	struct alignas(32) database_sample_record
	{
		sample_record record;
		uint64_t tier_metadata[2];
	};
	static_assert(sizeof(database_sample_record) == 32, "layout");

	// One per (segment, animated sub-track), 32 bytes: where the sub-track's bits sit inside a keyframe of that segment and
	// how to expand them, ready to use. Widths: 1..23 = quantized, 0 = constant in the segment (the 16 bit sample is pre-converted into
	// range_min, range_extent = 0, nothing is read), 32 / 33 / 34 = raw fp32 (ranges ignored; k_width_raw_* below).
	// Segment range values are float(u8) * (1/255) exactly as the reference computes them per pose
	// (animated_track_cache.transform.h:188-196, math/vector4_packing.h:781-818); single segment clips get min 0 / extent 1.
	// (Round 2 measured the compact alternative -- 8 byte entries expanded in-lane, bit offsets from a wavefront prefix sum over the
	// widths: it halves the L2 miss traffic of mixed-clip batches and costs ~58 VALU instructions per 64 sub-tracks, which these
	// VALU-issue-bound kernels pay in full: +11 % on one clip, +43 % on the 300-bone rig. profiles/r02_experiment_compact_plan_tables.txt)
	struct alignas(16) plan_entry
	{
		uint32_t bit_offset_and_width;	// bit offset inside the keyframe (low 24 bits) | num_bits << 24
		float inv_max_value;			// 1 / (2^num_bits - 1) (math/vector4_packing.h:927-935); 0 for width 0 (nothing is read), 1 for width 32
		float range_min[3];
		float range_extent[3];
	};

	// One per animated sub-track, 32 bytes: clip range (AOS copy of the blob's clip range data, which is SOA per group of 4
	// for rotations) and the track it belongs to.
	struct alignas(16) clip_range_entry
	{
		float range_min[3];
		uint32_t track_index;
		float range_extent[3];
		uint32_t quad_index;			// track_index * 3 + kind: where the sub-track lands in the pose
	};

	// Scalar track lists (float1f .. vector4f) get two tables, [num_tracks] each:
	//   scalar_track_header: where the track's bits sit inside a frame and their width (1..23 = quantized, 0 = constant: the value is
	//                        in the range row and nothing is read, 32 = raw fp32), and 1 / (2^num_bits - 1);
	//   a range row of 2 * C floats: min[C], extent[C] (constant tracks: the sample, 0; raw tracks: 0, 1).
	struct alignas(8) scalar_track_header
	{
		uint32_t bit_offset_and_width;	// bit offset from the blob start inside frame 0 (low 24 bits) | num_bits << 24
		float inv_max_value;			// math/scalar_packing.h:117-123
	};

	// One per (segment, pose window), 32 bytes, read through the scalar cache: where inside a keyframe of that segment the bits of the
	// window's animated sub-tracks sit. The bitstream orders a keyframe by kind (all rotations, all translations, all scales), each
	// in track order, so a window's sub-tracks form ONE contiguous run of bits per kind: [first_bit[k], end_bit[k]) from the start of
	// the keyframe (both 0 for a kind without bits in the window). The staged kernel copies these runs into LDS with coalesced 16 byte
	// reads and lanes pick their fields there, instead of 64 scattered unaligned loads per instruction from the bitstream itself.
	struct alignas(32) window_span_entry
	{
		uint32_t first_bit[3];
		uint32_t end_bit[3];
		uint32_t reserved[2];
	};
	static_assert(sizeof(window_span_entry) == 32, "layout");

	static_assert(sizeof(scalar_track_header) == 8, "layout");
	static_assert(sizeof(sample_record) == 16, "layout");
	static_assert(sizeof(plan_entry) == 32, "layout");
	static_assert(sizeof(clip_range_entry) == 32, "layout");

	// Animated sub-tracks are numbered by destination, not in bitstream order: the ones that land in a window of
	// k_image_chunk_quads consecutive quads form a contiguous range of ordinals, rotations first.
	// 312 = 104 whole tracks: a window never splits a track, and it holds an EVEN number of them -- what the compact output layouts
	// (32 / 40 bytes per track) and the re-tiling path need for windows that start 16 byte aligned (kernels_pose.inl); 4 992 bytes
	// also keep every 1 KiB store of a window on whole 64 byte granules of the write path (318 quads cost the rig 43 %)
#if !defined(ACLHIP_WINDOW_QUADS)
	#define ACLHIP_WINDOW_QUADS 312
#endif
	constexpr uint32_t k_image_chunk_quads = ACLHIP_WINDOW_QUADS;
	static_assert(k_image_chunk_quads % 6 == 0, "windows hold an even number of whole tracks (compact layouts: 16 byte aligned window starts)");

	__device__ __forceinline__ bool is_rotation_entry(const clip_range_entry& entry) { return entry.quad_index == entry.track_index * 3u; }

	// Markers in the W lane of a base pose quad: the bit patterns from 0xFFC00000 up (negative quiet NaNs). A real W is never one of
	// them: sqrt(|..|) for the drop-W rotation formats, 0 for vectors, and for quatf_full -- whose W is stored and may be NEGATIVE, which is
	// why the sign bit alone (rounds 1-5) no longer marks anything -- any float but a NaN (registration makes a garbage constant's NaN
	// positive; the device's own arithmetic produces 0x7FC00000).
	constexpr uint32_t k_quad_special = 0xFFC00000u;			// bits >= this: not a constant sub-track (is_special_quad)
	constexpr uint32_t k_quad_animated = 0x00200000u;			// special + this bit: low 21 bits = animated ordinal
	constexpr uint32_t k_quad_default_w_one = 0x00000001u;		// special, not animated: default sub-track, this bit = its identity W is 1
	constexpr uint32_t k_quad_ordinal_mask = 0x001FFFFFu;
	__host__ __device__ __forceinline__ bool is_special_quad(uint32_t w_bits) { return w_bits >= k_quad_special; }

	// Widths a plan_entry can name beyond the quantized 0..23 (bit_offset_and_width >> 24). The three raw classes store IEEE floats, big
	// endian, at any bit of the keyframe and skip both range expansions (animated_track_cache.transform.h:589-620,905-926):
	constexpr uint32_t k_width_raw_variable = 32u;		// the raw bit rate of a VARIABLE format: 3 floats; a rotation still passes through the SOA range
														// arithmetic as value * 1 + 0 twice (:316-349,420-465: a -0.0 comes out +0.0)
	constexpr uint32_t k_width_raw_full = 33u;			// quatf_drop_w_full / vector3f_full: 3 floats, no range arithmetic at all (:1384-1412 never runs)
	constexpr uint32_t k_width_raw_quat = 34u;			// quatf_full: 4 floats, W stored (unpack_vector4_128_unsafe, math/vector4_packing.h:59-164)
	__host__ __device__ __forceinline__ bool is_raw_width(uint32_t width) { return width >= k_width_raw_variable; }
	__host__ __device__ __forceinline__ uint32_t stored_sample_bits(uint32_t width) { return width < k_width_raw_variable ? width * 3u : (width == k_width_raw_quat ? 128u : 96u); }

	// Per clip record in HBM, written once at registration; read through the scalar cache by every wave. The FIRST 64 bytes hold what a
	// seek and a single track request need (k_clip_head_bytes: waves of mixed clips in decompress_track_kernel gather just these, four
	// lanes per record); the second half what only whole poses, databases and the table defaults read.
	// (Kernels that hold two or three records -- the pose consumers -- pass theirs through load_clip_fields, kernels_pose.inl: loaded as
	// two 16 register blocks, a record otherwise stays two blocks that live as long as any of their fields.)
	struct alignas(128) device_clip
	{
		const uint8_t* blob;					// the compressed_tracks bytes, unchanged, 16 byte aligned, >= 64 bytes of tail padding
		const float4* base_pose;				// [3 * num_tracks] rotation | translation | scale per track: constants expanded, defaults = identity, animated = marker
		const sample_record* samples;			// [num_samples]; database_sample_record[num_samples] for clips bound to a database (k_clip_database_samples)
		const clip_range_entry* clip_ranges;	// [num_animated]; scalar clips: float[num_tracks][2 * C] range rows
		uint32_t num_tracks;
		uint32_t num_samples;
		float sample_rate;
		float duration_clamp;					// calculate_finite_duration(num_samples)
		float duration_wrap;					// calculate_finite_duration(num_samples + 1)
		uint32_t flags;							// k_clip_*
		uint32_t num_segments;
		uint32_t num_animated;					// rotations + translations + scales; scalar clips: bits per frame
		// ---- 64 bytes ----
		const float4* resolved_pose;			// [3 * num_tracks] like base_pose but final: defaults hold the track_writer defaults, no markers
		const plan_entry* plan;					// [num_segments][num_animated]; scalar clips: scalar_track_header[num_tracks]
		const uint8_t* db_headers;				// database runtime clip/segment headers (device) or null
		const uint8_t* db_bulk_data[2];			// database bulk data, medium / low importance tier (device) or null
		const uint32_t* image_chunks;			// [ceil(3 * num_tracks / k_image_chunk_quads) + 1] first animated ordinal of every pose window; behind it,
												// at the next multiple of 32 bytes: window_span_entry[num_segments][num_windows] (window_spans_of)
		const uint32_t* hierarchy;				// aclhip_set_clip_hierarchy: walk schedules for 1 / 2 / 4 / 8 instances per workgroup; or null
		uint32_t db_clip_header_offset;			// into db_headers
		uint32_t reserved;
	};

	constexpr uint32_t k_clip_head_bytes = 64;
	static_assert(sizeof(device_clip) == 128 && offsetof(device_clip, num_animated) + 4 == k_clip_head_bytes && offsetof(device_clip, resolved_pose) == k_clip_head_bytes, "layout");

	constexpr uint32_t k_clip_has_scale = 1u << 0;
	constexpr uint32_t k_clip_has_stripped_keyframes = 1u << 1;	// stripped keyframes or database: sample_indices matter
	constexpr uint32_t k_clip_has_database = 1u << 2;
	constexpr uint32_t k_clip_wraps = 1u << 3;					// compressed_tracks::get_looping_policy() == wrap
	constexpr uint32_t k_clip_has_raw = 1u << 4;				// some (segment, sub-track) uses the raw bit rate
	constexpr uint32_t k_clip_is_scalar = 1u << 5;				// scalar track list: only the scalar kernel accepts it
	constexpr uint32_t k_clip_scaled = 1u << 6;					// scale sub-tracks, or a default scale other than 1: some scale of a pose may differ from 1
	constexpr uint32_t k_clip_database_samples = 1u << 7;		// bound to a database: `samples` holds database_sample_record (tier metadata copied per sample)
	constexpr uint32_t k_clip_components_shift = 8;				// scalar clips: floats per sample (1..4) in bits 8..10
	constexpr uint32_t k_clip_negative_scale = 1u << 11;			// some scale sub-track may decode a negative component: rtm::qvv_mul then composes matrices (pose consumers)
	constexpr uint32_t k_clip_short_exact_math = 1u << 12;		// no animated rotation of the clip can hand the kernels a square root argument in (0, 2^-96): the short exact forms apply (host_clips.inl)
	constexpr uint32_t k_clip_raw_rotations = 1u << 13;			// some rotation sub-track is stored raw (fp32) in some segment
	constexpr uint32_t k_clip_full_rotations = 1u << 14;			// rotation format quatf_full: W is stored, constant rotations are never normalized (constant_track_cache.transform.h:136-149)
	constexpr uint32_t k_clip_valid = 1u << 31;

	// pose windows of a transform clip, and where its window span table starts (behind image_chunks, 32 byte aligned)
	__host__ __device__ __forceinline__ uint32_t num_pose_windows(uint32_t num_tracks)
	{
		const uint32_t windows = (num_tracks * 3u + k_image_chunk_quads - 1u) / k_image_chunk_quads;
		return windows != 0 ? windows : 1u;
	}
	__host__ __device__ __forceinline__ uint32_t window_spans_word_offset(uint32_t num_windows) { return (num_windows + 1u + 7u) & ~7u; }

	// Behind a clip's resolved pose: the same pose as 10 packed floats per track (what the QVV40 layout starts from), then the clip's
	// BIND POSE, 12 floats per track (rotation xyzw | translation xyz 0 | scale xyz 0) -- track_desc_transformf::default_value of every
	// track from the blob's optional track descriptions, the identity when it carries none: what ACLHIP_DEFAULT_BIND_POSE resolves
	// default sub-tracks to. No pointer of its own: the 128 byte clip record is full.
	__host__ __device__ __forceinline__ uint32_t resolved_qvv40_floats(uint32_t num_tracks) { return (num_tracks * 10u + 3u) / 4u * 4u + 4u; }
	__device__ __forceinline__ const float* bind_pose_of(const device_clip& clip)
	{
		return reinterpret_cast<const float*>(clip.resolved_pose + size_t(clip.num_tracks) * 3u) + resolved_qvv40_floats(clip.num_tracks);
	}

	// LDS bytes that staging a run of `bits` keyframe bits takes: whole 16 byte pieces of the bitstream from the piece that holds the
	// run's first bit (the keyframe may start at any bit of any byte) to the piece that holds its last one, plus one piece of slack for
	// the 64 bit windows lanes read
	__host__ __device__ __forceinline__ uint32_t staged_run_bytes(uint32_t bits) { return ((bits + 126u) / 128u + 2u) * 16u; }

	// The transform kernels take valid transform clips, the scalar kernel valid scalar clips
	__device__ __forceinline__ bool is_transform_clip(uint32_t flags) { return (flags & (k_clip_valid | k_clip_is_scalar)) == k_clip_valid; }
	__device__ __forceinline__ bool is_scalar_clip(uint32_t flags) { return (flags & (k_clip_valid | k_clip_is_scalar)) == (k_clip_valid | k_clip_is_scalar); }

	// Launch wide settings (aclhip_decompress_params resolved to device pointers)
	struct decode_params
	{
		const float* default_values;
		const uint8_t* track_rounding_policies;
		const uint8_t* instance_rounding_policies;
		const uint32_t* instance_rows;	// pose kernels: row of the pose buffer each instance writes, or null (row = instance index)
		const uint32_t* time_indices;	// pose kernels: entry of sample_times each instance reads, or null (its own): instance lists kept in decode order
		const uint8_t* skip_tracks;		// compact pose kernels: per track, bit k set = its sub-track of kind k is not stored (aclhip_output_desc::skip_tracks), or null
		// per instance writer decisions (aclhip_output_desc, ABI 5) and looping policies; every per instance array is indexed by the
		// CALLER's instance index (caller_instance_of below)
		const uint8_t* mask_table;					// skip masks, mask_stride bytes each, or null
		const uint8_t* instance_masks;				// mask of every instance, or null
		const uint32_t* instance_track_counts;		// tracks every instance stores (its first K), or null
		const uint8_t* instance_looping_policies;	// or null
		const uint8_t* track_rounding_table;		// per instance writers' track rounding policies: tables of track_rounding_stride bytes, or null
		const uint8_t* instance_rounding_tables;	// table of every instance, or null
		uint32_t track_rounding_stride;
		uint32_t mask_stride;
		uint8_t rounding_policy;
		uint8_t looping_policy;
		uint8_t normalization;
		uint8_t per_track_rounding;
		uint8_t default_modes[3];
		uint8_t standard_defaults;		// 1 when default sub-tracks take the track_writer defaults (identity / zero / legacy scale) and normalization != always
		uint8_t standard_default_modes;	// 1 when default sub-tracks take the track_writer defaults, whatever the normalization policy
		uint8_t layout;					// aclhip_pose_layout (aclhip_output_desc)
		uint8_t skip_mask;				// bit k: sub-tracks of kind k (rotation / translation / scale) are not stored
		uint8_t items_per_wave;			// decompress_tracks_in_turn_kernel: work items a wave takes in turn
		uint8_t clips_by_caller_instance;	// pose kernels: 1 when the clip list is in the CALLER's instance order although the launch decodes in slot order (attached instance lists: clip = clips[time_indices[slot]])
		uint8_t fast_math;					// aclhip_decompress_params::flags & ACLHIP_DECODE_FAST (the host picks the kernels compiled for it)
		uint8_t user_defaults;				// some default sub-track takes a value from a table: default_values (constant / variable modes) or the clip's bind pose
	};

	// Per instance settings. `caller_instance`: the instance's index in the CALLER's lists -- instance lists decode in slot order and find
	// it in their order (decode_params::time_indices), everything else decodes instance i at slot i.
	__device__ __forceinline__ uint32_t instance_rounding_policy_of(const decode_params& params, uint32_t caller_instance)
	{
		return params.instance_rounding_policies != nullptr ? uint32_t(params.instance_rounding_policies[caller_instance]) : uint32_t(params.rounding_policy);
	}

	// decompression_context::set_looping_policy (decompress.h:149) belongs to one context = one instance
	__device__ __forceinline__ uint32_t instance_looping_policy_of(const decode_params& params, uint32_t caller_instance)
	{
		return params.instance_looping_policies != nullptr ? uint32_t(params.instance_looping_policies[caller_instance]) : uint32_t(params.looping_policy);
	}

	// The same two for a WAVE UNIFORM instance of the pose kernels, on the scalar unit: the dword that holds the instance's byte, always
	// requested -- from `always_readable` (any 4 byte aligned device address: the clip table) when the launch has no per instance array --
	// so that the request has no branch around it. Why not the byte itself: the scalar unit of gfx950 has no byte loads, a
	// global_load_ubyte + readfirstlane puts the byte on the VECTOR memory counter, and the wait the compiler places where the two paths
	// (array / no array) join is an s_waitcnt vmcnt(0) on BOTH: every wave of the pose kernels then waited there for its base pose DMA
	// -- 5 KiB through the texture unit, issued just before: 1.7 us of a 4.8 us life under the phase stamps (profiles/r05_experiments.md 8)
	// -- before its seek went on, whether or not the launch had such an array. (The pose consumers read their policies BEFORE any vector
	// memory operation and keep the byte loads: the scalar form costs them registers -- additive1 + object space 143.7 -> 165.9 us, measured.)
	__device__ __forceinline__ uint32_t uniform_instance_byte(const uint8_t* table, uint32_t index, const void* always_readable, uint32_t otherwise)
	{
		const uintptr_t address = table != nullptr ? reinterpret_cast<uintptr_t>(table) + index : reinterpret_cast<uintptr_t>(always_readable);
		const uint32_t word = *reinterpret_cast<const ACLHIP_CONSTANT uint32_t*>(address & ~uintptr_t(3));
		const uint32_t byte = (word >> ((uint32_t(address) & 3u) * 8u)) & 0xFFu;
		return table != nullptr ? byte : otherwise;
	}

	__device__ __forceinline__ uint32_t uniform_instance_rounding_policy_of(const decode_params& params, uint32_t caller_instance, const void* always_readable)
	{
		return uniform_instance_byte(params.instance_rounding_policies, caller_instance, always_readable, params.rounding_policy);
	}

	__device__ __forceinline__ uint32_t uniform_instance_looping_policy_of(const decode_params& params, uint32_t caller_instance, const void* always_readable)
	{
		return uniform_instance_byte(params.instance_looping_policies, caller_instance, always_readable, params.looping_policy);
	}

	// track_writer::get_rounding_policy(policy, track_index) (core/track_writer.h:97) belongs to the writer of ONE pose: the instance's own
	// table of per track policies, or the launch's
	__device__ __forceinline__ const uint8_t* instance_track_rounding_of(const decode_params& params, uint32_t caller_instance)
	{
		return params.instance_rounding_tables != nullptr
			? params.track_rounding_table + size_t(params.instance_rounding_tables[caller_instance]) * params.track_rounding_stride
			: params.track_rounding_policies;
	}

	// What happens to a decoded (local space) pose before it is stored (aclhip_pose_consumers resolved to device pointers)
	struct consumer_params
	{
		const uint32_t* base_clip_ids;		// [num_instances] the base clip instance each (additive) instance applies onto, or null
		const float* base_sample_times;
		const uint8_t* base_poses;			// precomputed base poses when base_clip_ids is null
		uint64_t base_pose_stride_bytes;
		uint32_t additive_format;			// acl::additive_clip_format8; 0 = no base
		uint32_t object_space;				// 1: local -> object space with the clip's hierarchy
		// blend of K clip instances (aclhip_pose_consumers::num_blend_clips): the K - 1 further clips and sample times of instance i at
		// [i * (K - 1) + j], its K weights at [i * K + k]
		const uint32_t* blend_clip_ids;
		const float* blend_sample_times;
		const float* blend_weights;
		uint32_t num_blend_clips;			// K; 0 / 1: no blend
	};

	// What seek leaves behind for the decode (persistent_transform_decompression_context_v0, decompression_context.transform.h:53-116)
	struct seek_state
	{
		const uint8_t* animated_track_data[2];	// first stored keyframe of each key's data source
		uint32_t segment_index[2];				// the two keys' segments: row of the plan each key follows
		uint32_t key_frame_bit_offsets[2];
		float interpolation_alpha;
		bool uses_single_segment;
	};

	// Output stores. Poses and values go to HBM once and these kernels never read them back: stored with system scope +
	// non-temporal hint (global_store ... sc0 sc1 nt) the write stream passes through the XCD's 4 MB L2 without evicting what the
	// decode keeps re-reading there -- clip records, table rows, keyframes. A batch that draws on 256 clips (30 MB of clip data)
	// takes 62 us instead of 94 us, a single-clip batch is unchanged (50 us); nt alone costs the single-clip batch 10 us, sc0 / sc1
	// alone change nothing. No builtin emits this combination for global stores, hence the inline assembly -- which hides
	// from the compiler that these ARE stores: a VMEM store of more than 8 bytes reads its data VGPRs for two more cycles on gfx940+
	// and a VALU write to them in that window corrupts what is stored (the compiler pads its own stores with s_nop 1; it cannot see
	// inside an asm). Every store below therefore carries its own `s_nop 1`. Found the hard way: the compact-output kernel reused a
	// data register one instruction after a store and wrote the loop counter into rotation.x under load.
	typedef float f32x4_store __attribute__((ext_vector_type(4)));
	typedef float f32x3_store __attribute__((ext_vector_type(3)));
	typedef float f32x2_store __attribute__((ext_vector_type(2)));

#if !defined(ACLHIP_STORE_MODIFIERS)
	#define ACLHIP_STORE_MODIFIERS "sc0 sc1 nt"
#endif
	__device__ __forceinline__ void store_streaming(void* address, f32x4_store value)
	{
		asm volatile("global_store_dwordx4 %0, %1, off " ACLHIP_STORE_MODIFIERS "\n\ts_nop 1" :: "v"(address), "v"(value) : "memory");
	}

	// C packed floats (4 byte aligned)
	template<uint32_t C>
	__device__ __forceinline__ void store_streaming_floats(float* address, const float (&value)[C])
	{
		static_assert(C >= 1 && C <= 4, "1 to 4 floats");
		if constexpr (C == 1)
			asm volatile("global_store_dword %0, %1, off sc0 sc1 nt" :: "v"(address), "v"(value[0]) : "memory");
		else if constexpr (C == 2)
			asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1 nt" :: "v"(address), "v"(f32x2_store{ value[0], value[1] }) : "memory");
		else if constexpr (C == 3)
			asm volatile("global_store_dwordx3 %0, %1, off sc0 sc1 nt\n\ts_nop 1" :: "v"(address), "v"(f32x3_store{ value[0], value[1], value[2] }) : "memory");
		else
			asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1 nt\n\ts_nop 1" :: "v"(address), "v"(f32x4_store{ value[0], value[1], value[2], value[3] }) : "memory");
	}

	// The same stores from a wave uniform base (SGPR pair) + a 32 bit byte offset per lane: no 64 bit address arithmetic per store
	template<uint32_t C>
	__device__ __forceinline__ void store_streaming_floats_at(float* uniform_base, uint32_t byte_offset, const float (&value)[C])
	{
		static_assert(C >= 1 && C <= 4, "1 to 4 floats");
		if constexpr (C == 1)
			asm volatile("global_store_dword %0, %1, %2 sc0 sc1 nt" :: "v"(byte_offset), "v"(value[0]), "s"(uniform_base) : "memory");
		else if constexpr (C == 2)
			asm volatile("global_store_dwordx2 %0, %1, %2 sc0 sc1 nt" :: "v"(byte_offset), "v"(f32x2_store{ value[0], value[1] }), "s"(uniform_base) : "memory");
		else if constexpr (C == 3)
			asm volatile("global_store_dwordx3 %0, %1, %2 sc0 sc1 nt\n\ts_nop 1" :: "v"(byte_offset), "v"(f32x3_store{ value[0], value[1], value[2] }), "s"(uniform_base) : "memory");
		else
			asm volatile("global_store_dwordx4 %0, %1, %2 sc0 sc1 nt\n\ts_nop 1" :: "v"(byte_offset), "v"(f32x4_store{ value[0], value[1], value[2], value[3] }), "s"(uniform_base) : "memory");
	}

	// One 16 byte sample record in a single (scalar, when the index is wave uniform) load
	typedef uint32_t u32x4_record __attribute__((ext_vector_type(4)));
	__device__ __forceinline__ sample_record load_sample_record(const sample_record* table, uint32_t index)
	{
		const u32x4_record raw = ((const ACLHIP_CONSTANT u32x4_record*)table)[index];
		sample_record record;
		__builtin_memcpy(&record, &raw, sizeof(record));
		return record;
	}

	typedef uint32_t u32x8_record __attribute__((ext_vector_type(8)));
	__device__ __forceinline__ database_sample_record load_database_sample_record(const sample_record* table, uint32_t index)
	{
		const u32x8_record raw = ((const ACLHIP_CONSTANT u32x8_record*)table)[index];
		database_sample_record record;
		__builtin_memcpy(&record, &raw, sizeof(record));
		return record;
	}

	// core/impl/interpolation_utils.impl.h:261-278
	__device__ __forceinline__ float apply_rounding_policy(float alpha, uint32_t policy)
	{
		if (policy == k_round_floor) return 0.0f;
		if (policy == k_round_ceil) return 1.0f;
		if (policy == k_round_nearest) return floorf(alpha + 0.5f);
		return alpha;
	}

	// core/impl/interpolation_utils.impl.h:224-253 with rounding_policy == none
	__device__ __forceinline__ float find_linear_interpolation_alpha(float sample_index, uint32_t index0, uint32_t index1)
	{
		if (index0 == index1)
			return 0.0f;
		if (index0 < index1)
			return (sample_index - float(index0)) / float(index1 - index0);
		return sample_index - float(index0);
	}

	// Clamp + find_linear_interpolation_samples_with_sample_rate (core/impl/interpolation_utils.impl.h:143-201): the two key frames
	// around a sample time and the (rounded) interpolation alpha. Shared by the transform and the scalar seek.
	__device__ __forceinline__ void find_key_frames(uint32_t clip_flags, uint32_t num_samples, float sample_rate, float duration_clamp, float duration_wrap,
		float sample_time, uint32_t rounding_policy, uint32_t looping_policy, uint32_t& out_key_frame0, uint32_t& out_key_frame1, float& out_alpha)
	{
		const bool wrap = looping_policy == k_loop_as_compressed ? (clip_flags & k_clip_wraps) != 0 : looping_policy == k_loop_wrap;
		const float clip_duration = wrap ? duration_wrap : duration_clamp;		// two values, then a select (not a select of addresses)

		// scalar_clamp (decompression.transform.h:215-216, decompression.scalar.h:189-190)
		sample_time = fminf(fmaxf(sample_time, 0.0f), clip_duration);

		const uint32_t last_sample_index = num_samples - 1;
		float sample_index = sample_time * sample_rate;
		uint32_t key_frame0 = uint32_t(sample_index);
		uint32_t key_frame1;
		if (!wrap)
			key_frame1 = min(key_frame0 + 1, last_sample_index);
		else if (key_frame0 > last_sample_index)
		{
			sample_index = 0.0f;
			key_frame0 = 0;
			key_frame1 = 0;
		}
		else
			key_frame1 = key_frame0 + 1 >= num_samples ? 0 : key_frame0 + 1;

		out_key_frame0 = key_frame0;
		out_key_frame1 = key_frame1;
		out_alpha = apply_rounding_policy(sample_index - float(key_frame0), rounding_policy);
	}

	// seek_v0 (decompression/impl/decompression.transform.h:206-563). Everything here is wave uniform in the pose kernel.
	// The reference guesses the segment and scans up to 4 start indices (:374-409); the lookup table gives the same answer.
	__device__ __forceinline__ void seek(const device_clip& clip, float sample_time, uint32_t rounding_policy, uint32_t looping_policy, seek_state& out)
	{
		uint32_t key_frame0, key_frame1;
		float alpha;
		find_key_frames(clip.flags, clip.num_samples, clip.sample_rate, clip.duration_clamp, clip.duration_wrap, sample_time, rounding_policy, looping_policy,
			key_frame0, key_frame1, alpha);

		// one scalar load per key: the sample's record, and for clips bound to a database the tier metadata of its segment with it
		const bool has_database = (clip.flags & k_clip_database_samples) != 0;
		sample_record segment0, segment1;
		uint64_t medium0 = 0, medium1 = 0, low0 = 0, low1 = 0;
		if (has_database)
		{
			const database_sample_record record0 = load_database_sample_record(clip.samples, key_frame0);
			const database_sample_record record1 = load_database_sample_record(clip.samples, key_frame1);
			segment0 = record0.record;
			segment1 = record1.record;
			medium0 = record0.tier_metadata[0]; low0 = record0.tier_metadata[1];
			medium1 = record1.tier_metadata[0]; low1 = record1.tier_metadata[1];
		}
		else
		{
			segment0 = load_sample_record(clip.samples, key_frame0);
			segment1 = load_sample_record(clip.samples, key_frame1);
		}
		const uint32_t segment_index0 = segment0.segment_and_local >> 5;
		const uint32_t segment_index1 = segment1.segment_and_local >> 5;

		uint32_t segment_key_frame0 = segment0.segment_and_local & 31u;
		uint32_t segment_key_frame1 = segment1.segment_and_local & 31u;
		const uint32_t start_index0 = key_frame0 - segment_key_frame0;
		const uint32_t start_index1 = key_frame1 - segment_key_frame1;

		const uint8_t* animated_track_data0 = clip.blob + segment0.animated_offset;
		const uint8_t* animated_track_data1 = clip.blob + segment1.animated_offset;

		if ((clip.flags & k_clip_has_stripped_keyframes) != 0)
		{
			// :272-362 / :411-515: snap to the nearest keyframes that are present, in the clip or in a streamed database tier
			uint32_t sample_indices0 = segment0.sample_indices;
			uint32_t sample_indices1 = segment1.sample_indices;
			const float clip_sample_index = alpha + float(key_frame0);
			if (has_database)
			{
				sample_indices0 |= uint32_t(medium0) | uint32_t(low0);
				sample_indices1 |= uint32_t(medium1) | uint32_t(low1);
			}

			const uint32_t candidate_indices0 = sample_indices0 & (0xFFFFFFFFu << (31 - segment_key_frame0));
			segment_key_frame0 = 31 - uint32_t(__builtin_ctz(candidate_indices0));
			const uint32_t candidate_indices1 = sample_indices1 & (0xFFFFFFFFu >> segment_key_frame1);
			segment_key_frame1 = uint32_t(__builtin_clz(candidate_indices1));

			alpha = find_linear_interpolation_alpha(clip_sample_index, start_index0 + segment_key_frame0, start_index1 + segment_key_frame1);

			sample_indices0 = segment0.sample_indices;
			sample_indices1 = segment1.sample_indices;

			if (has_database)
			{
				const uint64_t sample_bit0 = uint64_t(1) << (31 - segment_key_frame0);
				const uint64_t sample_bit1 = uint64_t(1) << (31 - segment_key_frame1);
				if ((medium0 & sample_bit0) != 0) { sample_indices0 = uint32_t(medium0); animated_track_data0 = clip.db_bulk_data[0] + uint32_t(medium0 >> 32); }
				else if ((low0 & sample_bit0) != 0) { sample_indices0 = uint32_t(low0); animated_track_data0 = clip.db_bulk_data[1] + uint32_t(low0 >> 32); }
				if ((medium1 & sample_bit1) != 0) { sample_indices1 = uint32_t(medium1); animated_track_data1 = clip.db_bulk_data[0] + uint32_t(medium1 >> 32); }
				else if ((low1 & sample_bit1) != 0) { sample_indices1 = uint32_t(low1); animated_track_data1 = clip.db_bulk_data[1] + uint32_t(low1 >> 32); }
			}

			// ordinal among the keyframes stored by the chosen data source
			segment_key_frame0 = uint32_t(__builtin_popcount(~(0xFFFFFFFFu >> segment_key_frame0) & sample_indices0));
			segment_key_frame1 = uint32_t(__builtin_popcount(~(0xFFFFFFFFu >> segment_key_frame1) & sample_indices1));
		}
		else
		{
			// every keyframe is stored: the sample's index inside its segment, which may be longer than 32 samples (sample_record)
			segment_key_frame0 = segment0.sample_indices;
			segment_key_frame1 = segment1.sample_indices;
		}

		// :530-562
		out.animated_track_data[0] = animated_track_data0;
		out.animated_track_data[1] = animated_track_data1;
		out.segment_index[0] = segment_index0;
		out.segment_index[1] = segment_index1;
		out.key_frame_bit_offsets[0] = segment_key_frame0 * segment0.pose_bit_size;
		out.key_frame_bit_offsets[1] = segment_key_frame1 * segment1.pose_bit_size;
		out.interpolation_alpha = alpha;
		out.uses_single_segment = segment_index0 == segment_index1;
	}

	// 4 / 8 bytes at any alignment (gfx950 global loads need no alignment)
	__device__ __forceinline__ uint32_t load_be32(const ACLHIP_CONSTANT uint8_t* p)
	{
		uint32_t v;
		__builtin_memcpy(&v, (const ACLHIP_CONSTANT void*)p, 4);
		return __builtin_bswap32(v);
	}

	__device__ __forceinline__ uint64_t load_u64(const ACLHIP_CONSTANT uint8_t* p)
	{
		uint64_t v;
		__builtin_memcpy(&v, (const ACLHIP_CONSTANT void*)p, 8);
		return v;
	}

	// Both keyframes of one animated sub-track, range expanded: the bit unpack of math/vector4_packing.h:921-1035 (quantized) and
	// :479-599 (raw), then the segment and clip range expansion of animated_track_cache.transform.h:302-350,391-466,930-960.
	// All four bitstream loads are issued before any of them is used. kHasRaw = false compiles the raw fix-up out.
	template<bool kHasRaw>
	__device__ __forceinline__ void unpack_animated_samples(const seek_state& state, const plan_entry& plan0, const plan_entry& plan1,
		const clip_range_entry& clip_range, bool is_rotation, float out_v0[3], float out_v1[3])
	{
		const ACLHIP_CONSTANT uint8_t* data0 = as_constant(state.animated_track_data[0]);
		const ACLHIP_CONSTANT uint8_t* data1 = as_constant(state.animated_track_data[1]);

		const uint32_t num_bits0 = plan0.bit_offset_and_width >> 24;
		const uint32_t num_bits1 = plan1.bit_offset_and_width >> 24;
		const uint32_t bit_offset0 = state.key_frame_bit_offsets[0] + (plan0.bit_offset_and_width & 0x00FFFFFFu);
		const uint32_t bit_offset1 = state.key_frame_bit_offsets[1] + (plan1.bit_offset_and_width & 0x00FFFFFFu);
		const uint32_t bit_offset_z0 = bit_offset0 + 2u * num_bits0;
		const uint32_t bit_offset_z1 = bit_offset1 + 2u * num_bits1;

		// x and y sit inside the 64 bit window that starts at the byte holding the first bit (7 + 2 * 23 <= 64); x even inside its
		// top 32 bits (7 + 23 <= 32). z gets its own 32 bit window.
		const uint64_t window_xy0 = load_u64(data0 + (bit_offset0 >> 3));
		const uint32_t window_z0 = load_be32(data0 + (bit_offset_z0 >> 3));
		const uint64_t window_xy1 = load_u64(data1 + (bit_offset1 >> 3));
		const uint32_t window_z1 = load_be32(data1 + (bit_offset_z1 >> 3));

		float v[2][3];
		#pragma unroll
		for (uint32_t key = 0; key < 2; ++key)
		{
			const uint64_t window_xy = key == 0 ? window_xy0 : window_xy1;
			const uint32_t hi_z = key == 0 ? window_z0 : window_z1;
			const uint32_t num_bits = key == 0 ? num_bits0 : num_bits1;
			const uint32_t shift_xy = (key == 0 ? bit_offset0 : bit_offset1) & 7u;
			const uint32_t shift_z = (key == 0 ? bit_offset_z0 : bit_offset_z1) & 7u;
			const plan_entry& plan = key == 0 ? plan0 : plan1;

			const uint32_t hi = __builtin_bswap32(uint32_t(window_xy));
			const uint32_t lo = __builtin_bswap32(uint32_t(window_xy >> 32));

			// v_bfe_u32: (source >> offset) & ((1 << width) - 1), and 0 for width 0 (a sub-track that is constant in its segment)
			const uint32_t x = __builtin_amdgcn_ubfe(hi, 32u - shift_xy - num_bits, num_bits);
			// bits [shift + w, shift + w + 32) of hi:lo; shift + w is in [1, 30] for real widths (a width 0 result is discarded by the bfe)
			const uint32_t window_y = __builtin_amdgcn_alignbit(hi, lo, 32u - (shift_xy + num_bits));
			const uint32_t y = __builtin_amdgcn_ubfe(window_y, 32u - num_bits, num_bits);
			const uint32_t z = __builtin_amdgcn_ubfe(hi_z, 32u - shift_z - num_bits, num_bits);

			const float quantized[3] = { float(x) * plan.inv_max_value, float(y) * plan.inv_max_value, float(z) * plan.inv_max_value };

			// v = v * segment_extent + segment_min, then v = v * clip_extent + clip_min (multiply, then add: never fused).
			// Constant-in-segment sub-tracks arrive here as 0 * 0 + sample; single segment clips as v * 1 + 0: exact for v >= +0.
			#pragma unroll
			for (uint32_t c = 0; c < 3; ++c)
			{
				// (also under ACLHIP_CONSUMERS_FAST: a rotation's W = sqrt(1 - x^2 - y^2 - z^2) turns one ulp of x, y or z into 1e-5 of W
				// for rotations near half a turn -- fusing these four operations, measured in round 4, moved rotations by up to 2.7e-5; the
				// fast mode therefore keeps x, y, z bit identical to the default's and is cheaper only behind them)
				const float segment_value = (quantized[c] * plan.range_extent[c]) + plan.range_min[c];
				v[key][c] = (segment_value * clip_range.range_extent[c]) + clip_range.range_min[c];
			}
		}

		if (kHasRaw)
		{
			// Raw (fp32) keyframes: three big endian floats starting at an arbitrary bit (math/vector4_packing.h:479-599). What the
			// code above computed for them is discarded. Raw samples skip both range expansions; in the reference's SOA rotation
			// path the ignored lanes still see value * 1 + 0 twice (animated_track_cache.transform.h:316-349,420-465), which only
			// matters for a -0.0.
			#pragma unroll
			for (uint32_t key = 0; key < 2; ++key)
			{
				const uint32_t key_bits = key == 0 ? num_bits0 : num_bits1;
				if (is_raw_width(key_bits))
				{
					const uint32_t bit_offset = key == 0 ? bit_offset0 : bit_offset1;
					const ACLHIP_CONSTANT uint8_t* bytes = (key == 0 ? data0 : data1) + (bit_offset >> 3);
					const uint32_t shift = bit_offset & 7u;
					const uint32_t w0 = load_be32(bytes), w1 = load_be32(bytes + 4), w2 = load_be32(bytes + 8), w3 = load_be32(bytes + 12);
					float raw[3] = { __uint_as_float(__funnelshift_l(w1, w0, shift)), __uint_as_float(__funnelshift_l(w2, w1, shift)), __uint_as_float(__funnelshift_l(w3, w2, shift)) };
					const bool through_ranges = is_rotation && key_bits == k_width_raw_variable;
					#pragma unroll
					for (uint32_t c = 0; c < 3; ++c)
						v[key][c] = through_ranges ? (((raw[c] * 1.0f) + 0.0f) * 1.0f) + 0.0f : raw[c];
				}
			}
		}

		#pragma unroll
		for (uint32_t c = 0; c < 3; ++c)
		{
			out_v0[c] = v[0][c];
			out_v1[c] = v[1][c];
		}
	}

	// The same unpack with ONE 16 byte load per key, from the DWORD that holds the sub-track's first bit: three components of up to 23
	// bits start at most 31 bits into it (100 bits), three raw floats end at bit 127. Round 3 measured what the texture unit makes of a
	// lane's reads of the bitstream: the 8 + 4 byte windows from the BYTE that holds the first bit (unpack_animated_samples above: two
	// requests per key at any byte address) cost the 300-bone rig 8 % of its launch against one dword aligned request per key, and the
	// texture unit is what that launch waits for (busy 83 - 93 %); the extraction costs ~30 more instructions and 8 more registers
	// per pass, which the one-window workloads -- bound by their write stream -- do not get back (64k x 100 bones: 49.6 -> 53.1 us). So:
	// the kernels of multi-window poses read this way (kWideKeyLoads), everything else as before. profiles/r03_experiments.md
	typedef uint32_t key_window __attribute__((ext_vector_type(4)));

	// the 32 bits that start `shift` (0..31) bits into the big endian pair hi:lo
	__device__ __forceinline__ uint32_t bits_from(uint32_t hi, uint32_t lo, uint32_t shift) { return __funnelshift_l(lo, hi, shift); }

	template<bool kHasRaw>
	__device__ __forceinline__ void unpack_animated_samples_wide(const seek_state& state, const plan_entry& plan0, const plan_entry& plan1,
		const clip_range_entry& clip_range, bool is_rotation, float out_v0[3], float out_v1[3])
	{
		const uint32_t num_bits0 = plan0.bit_offset_and_width >> 24;
		const uint32_t num_bits1 = plan1.bit_offset_and_width >> 24;
		const uint32_t bit_offset0 = state.key_frame_bit_offsets[0] + (plan0.bit_offset_and_width & 0x00FFFFFFu);
		const uint32_t bit_offset1 = state.key_frame_bit_offsets[1] + (plan1.bit_offset_and_width & 0x00FFFFFFu);

		// (the keyframe data is 4 byte aligned inside a blob -- compressed_headers.h:309-324 -- but a database tier's need not be: the dword is
		// found from the address itself)
		const uintptr_t byte_address0 = reinterpret_cast<uintptr_t>(state.animated_track_data[0]) + (bit_offset0 >> 3);
		const uintptr_t byte_address1 = reinterpret_cast<uintptr_t>(state.animated_track_data[1]) + (bit_offset1 >> 3);
		const key_window loaded0 = *reinterpret_cast<const ACLHIP_CONSTANT key_window*>(byte_address0 & ~uintptr_t(3));
		const key_window loaded1 = *reinterpret_cast<const ACLHIP_CONSTANT key_window*>(byte_address1 & ~uintptr_t(3));

		float v[2][3];
		#pragma unroll
		for (uint32_t key = 0; key < 2; ++key)
		{
			const key_window loaded = key == 0 ? loaded0 : loaded1;
			const uint32_t num_bits = key == 0 ? num_bits0 : num_bits1;
			const uint32_t shift_x = ((uint32_t(key == 0 ? byte_address0 : byte_address1) & 3u) << 3) | ((key == 0 ? bit_offset0 : bit_offset1) & 7u);		// 0..31
			const plan_entry& plan = key == 0 ? plan0 : plan1;

			// big endian dwords of the bitstream
			const uint32_t w0 = __builtin_bswap32(loaded.x), w1 = __builtin_bswap32(loaded.y), w2 = __builtin_bswap32(loaded.z), w3 = __builtin_bswap32(loaded.w);
			const uint32_t shift_y = shift_x + num_bits;			// 1..54: y starts in dword 0 or 1
			const uint32_t shift_z = shift_y + num_bits;			// 2..77: z starts in dword 0, 1 or 2
			const bool y_in_1 = shift_y >= 32u;
			const bool z_in_1 = shift_z >= 32u, z_in_2 = shift_z >= 64u;
			const uint32_t window_x = bits_from(w0, w1, shift_x);
			const uint32_t window_y = bits_from(y_in_1 ? w1 : w0, y_in_1 ? w2 : w1, shift_y & 31u);
			const uint32_t window_z = bits_from(z_in_2 ? w2 : (z_in_1 ? w1 : w0), z_in_2 ? w3 : (z_in_1 ? w2 : w1), shift_z & 31u);

			// v_bfe_u32: (source >> offset) & ((1 << width) - 1), and 0 for width 0 (a sub-track that is constant in its segment)
			const uint32_t x = __builtin_amdgcn_ubfe(window_x, 32u - num_bits, num_bits);
			const uint32_t y = __builtin_amdgcn_ubfe(window_y, 32u - num_bits, num_bits);
			const uint32_t z = __builtin_amdgcn_ubfe(window_z, 32u - num_bits, num_bits);

			const float quantized[3] = { float(x) * plan.inv_max_value, float(y) * plan.inv_max_value, float(z) * plan.inv_max_value };

			// v = v * segment_extent + segment_min, then v = v * clip_extent + clip_min (multiply, then add: never fused).
			// Constant-in-segment sub-tracks arrive here as 0 * 0 + sample; single segment clips as v * 1 + 0: exact for v >= +0.
			#pragma unroll
			for (uint32_t c = 0; c < 3; ++c)
			{
				const float segment_value = (quantized[c] * plan.range_extent[c]) + plan.range_min[c];
				v[key][c] = (segment_value * clip_range.range_extent[c]) + clip_range.range_min[c];
			}

			if (kHasRaw && is_raw_width(num_bits))
			{
				// Raw (fp32) keyframes: three big endian floats starting at an arbitrary bit (math/vector4_packing.h:479-599) -- the same
				// four dwords. What the code above computed for them is discarded. Raw samples skip both range expansions; in the
				// reference's SOA rotation path the ignored lanes still see value * 1 + 0 twice
				// (animated_track_cache.transform.h:316-349,420-465), which only matters for a -0.0.
				const float raw[3] = { __uint_as_float(bits_from(w0, w1, shift_x)), __uint_as_float(bits_from(w1, w2, shift_x)), __uint_as_float(bits_from(w2, w3, shift_x)) };
				const bool through_ranges = is_rotation && num_bits == k_width_raw_variable;
				#pragma unroll
				for (uint32_t c = 0; c < 3; ++c)
					v[key][c] = through_ranges ? (((raw[c] * 1.0f) + 0.0f) * 1.0f) + 0.0f : raw[c];
			}
		}

		#pragma unroll
		for (uint32_t c = 0; c < 3; ++c)
		{
			out_v0[c] = v[0][c];
			out_v1[c] = v[1][c];
		}
	}

#if defined(ACLHIP_EXPERIMENTS)
#include "../../tools/experiments/device_experiments.h"		// round 3's staged unpack: not part of a default build
#endif

	// ---- correctly rounded square root and reciprocal in fewer instructions than the compiler's general expansions (round 4) ------------
	// Every kernel but the headline's is bound by VALU issue, and 59 of a rotation's ~100 instructions were the compiler's
	// IEEE sqrtf (16 each: input scaling for tiny arguments, v_sqrt_f32, both neighbours tried against the residual, unscaling, a class
	// test) and 1.0f / x (11: v_div_scale twice, v_rcp_f32, refinement, v_div_fmas, v_div_fixup). The same neighbour test on the bare
	// v_sqrt_f32 (9 instructions) and two Newton steps on v_rcp_f32 (5) give the SAME BITS as sqrtf / 1.0f / x
	//   sqrt_rn_short   for x == +0 and 2^-96 <= x <= +inf      (below 2^-96 the residuals underflow: why the compiler scales)
	//   rcp_rn_short    for 2^-126 <= x <= 2^126                 (beyond, 1 / x is denormal or x is)
	// -- checked on every one of the 2^32 float bit patterns on the device (tools/probes/exact_math_probe.hip). Which form a wave takes is
	// a property of its CLIP, decided once at registration (k_clip_short_exact_math, host_clips.inl: no animated rotation of the clip can
	// produce an argument outside those ranges) and read from the clip record: a scalar branch, no per lane test.
	__device__ __forceinline__ float sqrt_rn_short(float x)
	{
		const float s = __builtin_amdgcn_sqrtf(x);		// within 1 ulp: the correctly rounded root is s or one of its neighbours
		const float down = __uint_as_float(__float_as_uint(s) - 1u), up = __uint_as_float(__float_as_uint(s) + 1u);
		const float residual_down = __builtin_fmaf(-down, s, x), residual_up = __builtin_fmaf(-up, s, x);
		float result = residual_down <= 0.0f ? down : s;
		result = residual_up > 0.0f ? up : result;
		return result;
	}

	__device__ __forceinline__ float rcp_rn_short(float x)
	{
		const float r0 = __builtin_amdgcn_rcpf(x);
		const float r1 = __builtin_fmaf(__builtin_fmaf(-x, r0, 1.0f), r0, r0);
		return __builtin_fmaf(__builtin_fmaf(-x, r1, 1.0f), r1, r1);
	}

	// math/quatf.h:135-147
	template<bool kShortExact = false>
	__device__ __forceinline__ float quat_from_positive_w(float x, float y, float z)
	{
		float w_squared = 1.0f - (x * x);
		w_squared = w_squared - (y * y);
		w_squared = w_squared - (z * z);
		return kShortExact ? sqrt_rn_short(fabsf(w_squared)) : sqrtf(fabsf(w_squared));
	}

	// math/quatf.h:200-211
	template<bool kShortExact = false>
	__device__ __forceinline__ float4 quat_normalize(float4 q)
	{
		float dot = q.x * q.x;
		dot = (q.y * q.y) + dot;
		dot = (q.z * q.z) + dot;
		dot = (q.w * q.w) + dot;
		const float inv_len = kShortExact ? rcp_rn_short(sqrt_rn_short(dot)) : 1.0f / sqrtf(dot);
		return make_float4(q.x * inv_len, q.y * inv_len, q.z * inv_len, q.w * inv_len);
	}

	// math/quatf.h:170-196
	__device__ __forceinline__ float4 quat_lerp_no_normalization(float4 q0, float4 q1, float alpha)
	{
		float dot = q0.x * q1.x;
		dot = (q0.y * q1.y) + dot;
		dot = (q0.z * q1.z) + dot;
		dot = (q0.w * q1.w) + dot;
		const uint32_t bias = __float_as_uint(dot) & 0x80000000u;
		float4 result;
		result.x = (__uint_as_float(__float_as_uint(q1.x) ^ bias) * alpha) + (q0.x - (q0.x * alpha));
		result.y = (__uint_as_float(__float_as_uint(q1.y) ^ bias) * alpha) + (q0.y - (q0.y * alpha));
		result.z = (__uint_as_float(__float_as_uint(q1.z) ^ bias) * alpha) + (q0.z - (q0.z * alpha));
		result.w = (__uint_as_float(__float_as_uint(q1.w) ^ bias) * alpha) + (q0.w - (q0.w * alpha));
		return result;
	}

	// rtm::vector_lerp in its stable form: end * alpha + (start - start * alpha)
	__device__ __forceinline__ float lerp_stable(float start, float end, float alpha) { return (end * alpha) + (start - (start * alpha)); }

	// ---- ACLHIP_CONSUMERS_FAST (aclhip_pose_consumers::flags): the same formulas in the hardware's cheapest correct-to-an-ulp forms ----
	// The default kernels follow the reference's x86 arithmetic one IEEE operation at a time and are bit exact with it; about half of a
	// rotation's instructions are then the expansions of a correctly rounded square root and division, and the pose consumers are bound
	// by instruction issue. Opt-in, per launch: v_sqrt_f32 / v_rsq_f32 (1 ulp) instead of the IEEE expansions, fused multiply-adds, and
	// quat_mul_vector3 as two cross products. Every result stays within a few ulp of the default's (tests/test_gpu_consumers.py holds
	// the poses to 2e-6 of the bit exact kernels and to the fp64 chain of test_pose_consumers_oracle.py).
	__device__ __forceinline__ float fast_sqrt(float x) { return __builtin_amdgcn_sqrtf(x); }
	__device__ __forceinline__ float fast_rsqrt(float x) { return __builtin_amdgcn_rsqf(x); }

	// (1 - x^2 - y^2 - z^2 cancels badly for rotations near half a turn: W then moves by 1e-6 and more with the rounding of the
	// squares, so those keep the reference's own operation order -- only the square root is the hardware's)
	__device__ __forceinline__ float quat_from_positive_w_fast(float x, float y, float z)
	{
		float w_squared = 1.0f - (x * x);
		w_squared = w_squared - (y * y);
		w_squared = w_squared - (z * z);
		return fast_sqrt(fabsf(w_squared));
	}

	__device__ __forceinline__ float4 quat_normalize_fast(float4 q)
	{
		const float dot = __builtin_fmaf(q.w, q.w, __builtin_fmaf(q.z, q.z, __builtin_fmaf(q.y, q.y, q.x * q.x)));
		const float inv_len = fast_rsqrt(dot);
		return make_float4(q.x * inv_len, q.y * inv_len, q.z * inv_len, q.w * inv_len);
	}

	__device__ __forceinline__ float4 quat_lerp_no_normalization_fast(float4 q0, float4 q1, float alpha)
	{
		const float dot = __builtin_fmaf(q0.w, q1.w, __builtin_fmaf(q0.z, q1.z, __builtin_fmaf(q0.y, q1.y, q0.x * q1.x)));
		const float signed_alpha = __uint_as_float(__float_as_uint(alpha) ^ (__float_as_uint(dot) & 0x80000000u));
		const float beta = 1.0f - alpha;
		return make_float4(__builtin_fmaf(q1.x, signed_alpha, q0.x * beta), __builtin_fmaf(q1.y, signed_alpha, q0.y * beta),
			__builtin_fmaf(q1.z, signed_alpha, q0.z * beta), __builtin_fmaf(q1.w, signed_alpha, q0.w * beta));
	}

	// rtm::quat_mul's formula (lhs first, then rhs), one multiply and three fused multiply-adds per lane
	__device__ __forceinline__ float4 quat_mul_fast(float4 l, float4 r)
	{
		float4 result;
		result.x = __builtin_fmaf(r.w, l.x, __builtin_fmaf(r.x, l.w, __builtin_fmaf(r.y, l.z, -(r.z * l.y))));
		result.y = __builtin_fmaf(r.w, l.y, __builtin_fmaf(-r.x, l.z, __builtin_fmaf(r.y, l.w, r.z * l.x)));
		result.z = __builtin_fmaf(r.w, l.z, __builtin_fmaf(r.x, l.y, __builtin_fmaf(-r.y, l.x, r.z * l.w)));
		result.w = __builtin_fmaf(r.w, l.w, __builtin_fmaf(-r.x, l.x, __builtin_fmaf(-r.y, l.y, -(r.z * l.z))));
		return result;
	}

	// quat_mul_vector3(v, q) = quat_mul(quat_mul(conj(q), (v, 0)), q) -- in Hamilton's notation q (v, 0) q*, rtm::quat_mul(lhs, rhs)
	// being rhs x lhs -- as v + w t + u x t with u = q.xyz, t = 2 (u x v): two cross products instead of two quaternion products
	__device__ __forceinline__ float4 quat_mul_vector3_fast(float4 v, float4 q)
	{
		const float ux = q.x, uy = q.y, uz = q.z;
		float tx = __builtin_fmaf(uy, v.z, -(uz * v.y)), ty = __builtin_fmaf(uz, v.x, -(ux * v.z)), tz = __builtin_fmaf(ux, v.y, -(uy * v.x));
		tx += tx; ty += ty; tz += tz;
		const float cx = __builtin_fmaf(uy, tz, -(uz * ty)), cy = __builtin_fmaf(uz, tx, -(ux * tz)), cz = __builtin_fmaf(ux, ty, -(uy * tx));
		return make_float4(__builtin_fmaf(q.w, tx, v.x) + cx, __builtin_fmaf(q.w, ty, v.y) + cy, __builtin_fmaf(q.w, tz, v.z) + cz, 0.0f);
	}

	// Decodes one animated sub-track of one instance: both keyframes, range expansion, W reconstruction, interpolation.
	// `policy` is the effective rounding policy of the track (none unless per track rounding is enabled);
	// `lerp_alpha` the alpha handed to the interpolation.
	// kHasRaw = false compiles the raw bit rate out, kPolicies = false the per track rounding policies.
	// kFastMath: 0 = the reference's arithmetic, bit for bit; 1 = ACLHIP_CONSUMERS_FAST (rotations and vectors in the hardware's cheapest
	// forms); 2 = ACLHIP_DECODE_FAST (rotations only: translations and scales stay bit identical)
	template<bool kPolicies, uint32_t kFastMath = 0>
	__device__ __forceinline__ float4 interpolate_animated_samples(const seek_state& state, const float (&v0)[3], const float (&v1)[3],
		bool is_rotation, uint32_t policy, float lerp_alpha, uint32_t normalization, bool normalize_samples, bool short_exact_math);

	// short_exact_math (WAVE UNIFORM): the clip's k_clip_short_exact_math -- its rotations may take sqrt_rn_short / rcp_rn_short
	template<bool kHasRaw, bool kPolicies, bool kWideKeyLoads = false, uint32_t kFastMath = 0>
	__device__ __forceinline__ float4 decode_animated_sub_track(const seek_state& state, const plan_entry& plan0, const plan_entry& plan1,
		const clip_range_entry& clip_range, bool is_rotation, uint32_t policy, float lerp_alpha, uint32_t normalization, bool normalize_samples, bool short_exact_math = false)
	{
		float v0[3], v1[3];
		if constexpr (kWideKeyLoads)
			unpack_animated_samples_wide<kHasRaw>(state, plan0, plan1, clip_range, is_rotation, v0, v1);
		else
			unpack_animated_samples<kHasRaw>(state, plan0, plan1, clip_range, is_rotation, v0, v1);
		if constexpr (kHasRaw)
		{
			// quatf_full: the sample's W is the fourth float of its 128 bits (unpack_vector4_128_unsafe) -- no reconstruction, no sample
			// normalization (animated_track_cache.transform.h:1416-1475 covers the drop-W formats only); a clip's rotations all share the
			// format, so both keys are of this class or neither is
			const bool stored_w = (plan0.bit_offset_and_width >> 24) == k_width_raw_quat;
			if (__builtin_amdgcn_ballot_w64(stored_w) != 0)
			{
				float w0 = 0.0f, w1 = 0.0f;
				if (stored_w)
				{
					const uint32_t bit_offset0 = state.key_frame_bit_offsets[0] + (plan0.bit_offset_and_width & 0x00FFFFFFu) + 96u;
					const uint32_t bit_offset1 = state.key_frame_bit_offsets[1] + (plan1.bit_offset_and_width & 0x00FFFFFFu) + 96u;
					const ACLHIP_CONSTANT uint8_t* bytes0 = as_constant(state.animated_track_data[0]) + (bit_offset0 >> 3);
					const ACLHIP_CONSTANT uint8_t* bytes1 = as_constant(state.animated_track_data[1]) + (bit_offset1 >> 3);
					w0 = __uint_as_float(__funnelshift_l(load_be32(bytes0 + 4), load_be32(bytes0), bit_offset0 & 7u));
					w1 = __uint_as_float(__funnelshift_l(load_be32(bytes1 + 4), load_be32(bytes1), bit_offset1 & 7u));
				}
				// (the wave's other lanes -- the vectors of test_data/configs/uniformly_sampled_mixed_var_0, say -- take their usual route)
				if (is_rotation)
					return interpolate_animated_rotation<kPolicies, false, true>(state, v0, v1, policy, lerp_alpha, normalization, normalize_samples, stored_w, w0, w1);
				return interpolate_animated_samples<kPolicies, kFastMath>(state, v0, v1, false, policy, lerp_alpha, normalization, normalize_samples, false);
			}
		}
		// (raw samples are any floats: a wave that meets one keeps the compiler's forms)
		return interpolate_animated_samples<kPolicies, kFastMath>(state, v0, v1, is_rotation, policy, lerp_alpha, normalization, normalize_samples, !kHasRaw && short_exact_math);
	}

	// What follows the unpack: W reconstruction, interpolation, normalization (rotations) / the stable lerp (translations, scales)
	// The rotation arithmetic of interpolate_animated_samples, with the compiler's or the short exact square roots / reciprocal
	// kStoredW (quatf_full, lanes with stored_w set): the samples carry their W (w0, w1) -- no reconstruction and no sample normalization,
	// which animated_track_cache.transform.h:1416-1475 applies to the drop-W formats only. should_interpolate_samples
	// (decompression_context.transform.h:192-199) is true for every settings type that supports more than one rotation format, the only
	// kind this library stands in for: the samples are always interpolated, then normalized under lerp_only / always.
	template<bool kPolicies, bool kShortExact, bool kStoredW = false>
	__device__ __forceinline__ float4 interpolate_animated_rotation(const seek_state& state, const float (&v0)[3], const float (&v1)[3],
		uint32_t policy, float lerp_alpha, uint32_t normalization, bool normalize_samples, bool stored_w = false, float w0 = 0.0f, float w1 = 0.0f)
	{
		float4 q0 = make_float4(v0[0], v0[1], v0[2], quat_from_positive_w<kShortExact>(v0[0], v0[1], v0[2]));
		float4 q1 = make_float4(v1[0], v1[1], v1[2], quat_from_positive_w<kShortExact>(v1[0], v1[1], v1[2]));
		if (kStoredW)
		{
			q0.w = stored_w ? w0 : q0.w;
			q1.w = stored_w ? w1 : q1.w;
		}

		// animated_track_cache.transform.h:1463-1473
		if (kPolicies && normalize_samples && !(kStoredW && stored_w))
		{
			q0 = quat_normalize<kShortExact>(q0);
			q1 = quat_normalize<kShortExact>(q1);
		}

		if (kPolicies)
		{
			if (policy == k_round_floor)
				return q0;
			if (policy == k_round_ceil)
				return q1;
			if (policy == k_round_nearest)
				return state.interpolation_alpha < 0.5f ? q0 : q1;
		}

		// :1604-1616
		float4 result = quat_lerp_no_normalization(q0, q1, lerp_alpha);
		if (normalization >= 1)
			result = quat_normalize<kShortExact>(result);
		return result;
	}

	template<bool kPolicies, uint32_t kFastMath>
	__device__ __forceinline__ float4 interpolate_animated_samples(const seek_state& state, const float (&v0)[3], const float (&v1)[3],
		bool is_rotation, uint32_t policy, float lerp_alpha, uint32_t normalization, bool normalize_samples, bool short_exact_math)
	{
		if constexpr (kFastMath != 0 && !kPolicies)
		{
			// ACLHIP_CONSUMERS_FAST: the same formulas, 1 ulp square roots and fused multiply-adds (see quat_normalize_fast above)
			if (is_rotation && __builtin_amdgcn_ballot_w64(is_rotation) != 0)
			{
				const float4 q0 = make_float4(v0[0], v0[1], v0[2], quat_from_positive_w_fast(v0[0], v0[1], v0[2]));
				const float4 q1 = make_float4(v1[0], v1[1], v1[2], quat_from_positive_w_fast(v1[0], v1[1], v1[2]));
				float4 result = quat_lerp_no_normalization_fast(q0, q1, lerp_alpha);
				if (normalization >= 1)
					result = quat_normalize_fast(result);
				return result;
			}
			// ACLHIP_DECODE_FAST keeps translations and scales exact: their error would scale with the rig's size, a rotation's does not
			if constexpr (kFastMath == 1)
			{
				const float beta = 1.0f - lerp_alpha;
				return make_float4(__builtin_fmaf(v1[0], lerp_alpha, v0[0] * beta), __builtin_fmaf(v1[1], lerp_alpha, v0[1] * beta), __builtin_fmaf(v1[2], lerp_alpha, v0[2] * beta), 0.0f);
			}
			return make_float4(lerp_stable(v0[0], v1[0], lerp_alpha), lerp_stable(v0[1], v1[1], lerp_alpha), lerp_stable(v0[2], v1[2], lerp_alpha), 0.0f);
		}

		// (the rotation arithmetic -- three square roots and a division, ~90 instructions -- sits behind a branch the WAVE takes: the
		// compiler otherwise turns `if (is_rotation)` into selects, and a pass of translations and scales pays for rotations it does not have)
		if (is_rotation && __builtin_amdgcn_ballot_w64(is_rotation) != 0)
		{
			// (a scalar branch: short_exact_math is the clip's flag, the same in every lane)
			if (short_exact_math)
				return interpolate_animated_rotation<kPolicies, true>(state, v0, v1, policy, lerp_alpha, normalization, normalize_samples);
			return interpolate_animated_rotation<kPolicies, false>(state, v0, v1, policy, lerp_alpha, normalization, normalize_samples);
		}

		// unpack_translation_group / unpack_scale_group, animated_track_cache.transform.h:1774-1836,1896-1958
		if (kPolicies)
		{
			if (policy == k_round_floor)
				return make_float4(v0[0], v0[1], v0[2], 0.0f);
			if (policy == k_round_ceil)
				return make_float4(v1[0], v1[1], v1[2], 0.0f);
			if (policy == k_round_nearest)
				return state.interpolation_alpha < 0.5f ? make_float4(v0[0], v0[1], v0[2], 0.0f) : make_float4(v1[0], v1[1], v1[2], 0.0f);
		}

		return make_float4(lerp_stable(v0[0], v1[0], lerp_alpha), lerp_stable(v0[1], v1[1], lerp_alpha), lerp_stable(v0[2], v1[2], lerp_alpha), 0.0f);
	}

	// ---- pose consumers (core/additive_utils.h:128-160, compression/transform_pose_utils.h:35-50) ----------------------------------
	// The reference writes these in Realtime Math (rtm::quat_mul / qvv_mul / qvv_normalize), a submodule that is absent from its
	// checkout: restated from RTM's x86 forms -- every lane of quat_mul sums its four products pairwise, (a + b) + (c + d), signs
	// folded into the products -- with two documented differences: the rotation is normalized with the decoder's
	// own correctly rounded sqrt + division (RTM starts from the hardware rsqrt ESTIMATE, which is not reproducible between CPUs),
	// and negative scales do not take RTM's detour through a matrix.
	struct qvv
	{
		float4 rotation;
		float4 translation;		// w = 0
		float4 scale;			// w = 0
	};

	// rtm::quat_mul, x86 form: every lane sums its four products pairwise, signs folded into the products:
	//   x = ((rw * lx) +  (rx * lw)) + ( (ry * lz) + -(rz * ly))
	//   y = ((rw * ly) + -(rx * lz)) + ( (ry * lw) +  (rz * lx))
	//   z = ((rw * lz) +  (rx * ly)) + (-(ry * lx) +  (rz * lw))
	//   w = ((rw * lw) + -(rx * lx)) + (-(ry * ly) + -(rz * lz))
	// Written as 8 v_pk_mul_f32 + 6 v_pk_add_f32 on the register pairs (x, y) and (z, w): the swizzles, broadcasts and signs are the
	// instructions' own op_sel / neg modifiers (negating an operand negates the product exactly; the sums keep the order above). The
	// compiler's version of the same C++ spends 10 more v_mov_b32 per product on building swizzled pairs -- and the object space walk
	// is three products per transform on a kernel that is bound by VALU issue.
	// A packed fp32 result may be read two instructions later at the earliest (gfx940+; the compiler pads its own instructions but
	// does not look inside an asm block: the order below keeps that distance, the s_nop in front and behind cover the block's edges).
	typedef float f32x2_lanes __attribute__((ext_vector_type(2)));
	// kConjugateLhs: the product of conjugate(lhs) and rhs -- the signs of lx, ly, lz fold into the same modifiers
	template<bool kConjugateLhs>
	__device__ __forceinline__ float4 quat_mul_packed(float4 lhs, float4 rhs)
	{
		const f32x2_lanes l01 = { lhs.x, lhs.y }, l23 = { lhs.z, lhs.w }, r01 = { rhs.x, rhs.y }, r23 = { rhs.z, rhs.w };
		f32x2_lanes t1, t2, t3, t4, u1, u2, u3, u4, xy, zw;
		if constexpr (!kConjugateLhs)
			asm("s_nop 0\n\t"																				// (an operand may come straight out of a packed instruction of the compiler's)
				"v_pk_mul_f32 %0, %13, %10 op_sel:[1,0] op_sel_hi:[1,1]\n\t"								// t1 = { rw * lx,  rw * ly }
				"v_pk_mul_f32 %1, %12, %11 op_sel:[0,1] op_sel_hi:[0,0] neg_lo:[0,0] neg_hi:[1,0]\n\t"		// t2 = { rx * lw, -rx * lz }
				"v_pk_mul_f32 %2, %12, %11 op_sel:[1,0] op_sel_hi:[1,1]\n\t"								// t3 = { ry * lz,  ry * lw }
				"v_pk_mul_f32 %3, %13, %10 op_sel:[0,1] op_sel_hi:[0,0] neg_lo:[1,0] neg_hi:[0,0]\n\t"		// t4 = { -rz * ly, rz * lx }
				"v_pk_mul_f32 %4, %13, %11 op_sel:[1,0] op_sel_hi:[1,1]\n\t"								// u1 = { rw * lz,  rw * lw }
				"v_pk_mul_f32 %5, %12, %10 op_sel:[0,1] op_sel_hi:[0,0] neg_lo:[0,0] neg_hi:[1,0]\n\t"		// u2 = { rx * ly, -rx * lx }
				"v_pk_mul_f32 %6, %12, %10 op_sel:[1,0] op_sel_hi:[1,1] neg_lo:[1,0] neg_hi:[1,0]\n\t"		// u3 = { -ry * lx, -ry * ly }
				"v_pk_mul_f32 %7, %13, %11 op_sel:[0,1] op_sel_hi:[0,0] neg_lo:[0,0] neg_hi:[1,0]\n\t"		// u4 = { rz * lw, -rz * lz }
				"v_pk_add_f32 %0, %0, %1\n\t"
				"v_pk_add_f32 %2, %2, %3\n\t"
				"v_pk_add_f32 %4, %4, %5\n\t"
				"v_pk_add_f32 %6, %6, %7\n\t"
				"v_pk_add_f32 %8, %0, %2\n\t"
				"v_pk_add_f32 %9, %4, %6\n\t"
				"s_nop 0"
				: "=&v"(t1), "=&v"(t2), "=&v"(t3), "=&v"(t4), "=&v"(u1), "=&v"(u2), "=&v"(u3), "=&v"(u4), "=&v"(xy), "=&v"(zw)
				: "v"(l01), "v"(l23), "v"(r01), "v"(r23));
		else
			asm("s_nop 0\n\t"
				"v_pk_mul_f32 %0, %13, %10 op_sel:[1,0] op_sel_hi:[1,1] neg_lo:[1,0] neg_hi:[1,0]\n\t"		// t1 = { rw * -lx,  rw * -ly }
				"v_pk_mul_f32 %1, %12, %11 op_sel:[0,1] op_sel_hi:[0,0]\n\t"								// t2 = { rx * lw, -rx * -lz }
				"v_pk_mul_f32 %2, %12, %11 op_sel:[1,0] op_sel_hi:[1,1] neg_lo:[1,0] neg_hi:[0,0]\n\t"		// t3 = { ry * -lz,  ry * lw }
				"v_pk_mul_f32 %3, %13, %10 op_sel:[0,1] op_sel_hi:[0,0] neg_lo:[0,0] neg_hi:[1,0]\n\t"		// t4 = { -rz * -ly, rz * -lx }
				"v_pk_mul_f32 %4, %13, %11 op_sel:[1,0] op_sel_hi:[1,1] neg_lo:[1,0] neg_hi:[0,0]\n\t"		// u1 = { rw * -lz,  rw * lw }
				"v_pk_mul_f32 %5, %12, %10 op_sel:[0,1] op_sel_hi:[0,0] neg_lo:[1,0] neg_hi:[0,0]\n\t"		// u2 = { rx * -ly, -rx * -lx }
				"v_pk_mul_f32 %6, %12, %10 op_sel:[1,0] op_sel_hi:[1,1]\n\t"								// u3 = { -ry * -lx, -ry * -ly }
				"v_pk_mul_f32 %7, %13, %11 op_sel:[0,1] op_sel_hi:[0,0]\n\t"								// u4 = { rz * lw, -rz * -lz }
				"v_pk_add_f32 %0, %0, %1\n\t"
				"v_pk_add_f32 %2, %2, %3\n\t"
				"v_pk_add_f32 %4, %4, %5\n\t"
				"v_pk_add_f32 %6, %6, %7\n\t"
				"v_pk_add_f32 %8, %0, %2\n\t"
				"v_pk_add_f32 %9, %4, %6\n\t"
				"s_nop 0"
				: "=&v"(t1), "=&v"(t2), "=&v"(t3), "=&v"(t4), "=&v"(u1), "=&v"(u2), "=&v"(u3), "=&v"(u4), "=&v"(xy), "=&v"(zw)
				: "v"(l01), "v"(l23), "v"(r01), "v"(r23));
		return make_float4(xy.x, xy.y, zw.x, zw.y);
	}

	__device__ __forceinline__ float4 quat_mul(float4 lhs, float4 rhs)
	{
		return quat_mul_packed<false>(lhs, rhs);
	}

	// quat_mul(quat_mul(conjugate(rotation), (vector.xyz, 0)), rotation); the W lane of the result is numeric residue and dropped
	__device__ __forceinline__ float4 quat_mul_vector3(float4 vector, float4 rotation)
	{
		const float4 vector_quat = make_float4(vector.x, vector.y, vector.z, 0.0f);
		const float4 result = quat_mul(quat_mul_packed<true>(rotation, vector_quat), rotation);
		return make_float4(result.x, result.y, result.z, 0.0f);
	}

	// rtm::qvv_mul leaves the quaternion path when any scale component of either side is negative (mirrored rigs: a quaternion cannot
	// carry a reflection) and composes 3x4 matrices instead. The kernels do the same (qvv_mul_through_matrices below) for exactly
	// those transforms, and keep COUNTING them (aclhip_get_negative_scale_count).
	__device__ __forceinline__ bool qvv_mul_takes_matrix_path(const qvv& lhs, const qvv& rhs)
	{
		return fminf(fminf(fminf(lhs.scale.x, lhs.scale.y), lhs.scale.z), fminf(fminf(rhs.scale.x, rhs.scale.y), rhs.scale.z)) < 0.0f;
	}

	// RTM 2.x's route for negative scales, restated operation by operation (the reciprocal square roots are the correctly rounded
	// 1 / sqrt like everywhere else in this file; pinned by tests/test_gpu_consumers.py against an fp64 matrix chain):
	//   matrix_from_qvv(lhs) * matrix_from_qvv(rhs)   row vectors, lhs first; every row ((x * X + y * Y) + z * Z) [+ W]
	//   matrix_remove_scale                            every axis normalized by its own length (left alone below a squared length of 1e-8)
	//   axis * sign(lhs.scale * rhs.scale)             +1 / -1 per axis (+1 for zero)
	//   rotation = quat_from_matrix(axes), normalized; translation = the product's W row; scale = lhs.scale * rhs.scale
	// Rare (mirrored rigs only): kept out of line so that the walk's registers are those of the quaternion path.
	struct matrix3x3_rows { float m[3][3]; };

	__device__ __forceinline__ matrix3x3_rows rotation_scale_matrix_of(const qvv& t)
	{
		const float x = t.rotation.x, y = t.rotation.y, z = t.rotation.z, w = t.rotation.w;
		const float x2 = x + x, y2 = y + y, z2 = z + z;
		const float xx = x * x2, xy = x * y2, xz = x * z2, yy = y * y2, yz = y * z2, zz = z * z2, wx = w * x2, wy = w * y2, wz = w * z2;
		matrix3x3_rows r;
		r.m[0][0] = (1.0f - (yy + zz)) * t.scale.x; r.m[0][1] = (xy + wz) * t.scale.x; r.m[0][2] = (xz - wy) * t.scale.x;
		r.m[1][0] = (xy - wz) * t.scale.y; r.m[1][1] = (1.0f - (xx + zz)) * t.scale.y; r.m[1][2] = (yz + wx) * t.scale.y;
		r.m[2][0] = (xz + wy) * t.scale.z; r.m[2][1] = (yz - wx) * t.scale.z; r.m[2][2] = (1.0f - (xx + yy)) * t.scale.z;
		return r;
	}

	__device__ __forceinline__ qvv qvv_mul_through_matrices(const qvv& lhs, const qvv& rhs)
	{
		const matrix3x3_rows l = rotation_scale_matrix_of(lhs);
		const matrix3x3_rows r = rotation_scale_matrix_of(rhs);
		const float rhs_translation[3] = { rhs.translation.x, rhs.translation.y, rhs.translation.z };
		const float lhs_translation[3] = { lhs.translation.x, lhs.translation.y, lhs.translation.z };
		const float scale[3] = { lhs.scale.x * rhs.scale.x, lhs.scale.y * rhs.scale.y, lhs.scale.z * rhs.scale.z };

		float axes[3][3], translation[3];
		#pragma unroll
		for (uint32_t c = 0; c < 3; ++c)
			translation[c] = rhs_translation[c] + (((lhs_translation[0] * r.m[0][c]) + (lhs_translation[1] * r.m[1][c])) + (lhs_translation[2] * r.m[2][c]));
		#pragma unroll
		for (uint32_t row = 0; row < 3; ++row)
		{
			float axis[3];
			#pragma unroll
			for (uint32_t c = 0; c < 3; ++c)
				axis[c] = ((l.m[row][0] * r.m[0][c]) + (l.m[row][1] * r.m[1][c])) + (l.m[row][2] * r.m[2][c]);
			const float length_squared = ((axis[0] * axis[0]) + (axis[1] * axis[1])) + (axis[2] * axis[2]);
			const float inv_length = length_squared >= 1.0e-8f ? 1.0f / sqrtf(length_squared) : 1.0f;
			const float sign = scale[row] >= 0.0f ? 1.0f : -1.0f;
			#pragma unroll
			for (uint32_t c = 0; c < 3; ++c)
				axes[row][c] = (length_squared >= 1.0e-8f ? axis[c] * inv_length : axis[c]) * sign;
		}

		// rtm::quat_from_matrix
		float q[4];
		const float trace = (axes[0][0] + axes[1][1]) + axes[2][2];
		if (trace > 0.0f)
		{
			const float inv_trace = 1.0f / sqrtf(trace + 1.0f);
			const float half_inv_trace = inv_trace * 0.5f;
			q[0] = (axes[1][2] - axes[2][1]) * half_inv_trace;
			q[1] = (axes[2][0] - axes[0][2]) * half_inv_trace;
			q[2] = (axes[0][1] - axes[1][0]) * half_inv_trace;
			q[3] = (1.0f / inv_trace) * 0.5f;
		}
		else
		{
			// the largest diagonal element leads; written per case so that every index is a constant (no scratch memory)
			const bool y_leads = axes[1][1] > axes[0][0];
			const bool z_leads = axes[2][2] > (y_leads ? axes[1][1] : axes[0][0]);
			const uint32_t best = z_leads ? 2u : (y_leads ? 1u : 0u);
			float best_best, next_next, last_last, best_next, next_best, best_last, last_best, next_last, last_next;
			if (best == 0)
			{
				best_best = axes[0][0]; next_next = axes[1][1]; last_last = axes[2][2];
				best_next = axes[0][1]; next_best = axes[1][0]; best_last = axes[0][2]; last_best = axes[2][0]; next_last = axes[1][2]; last_next = axes[2][1];
			}
			else if (best == 1)
			{
				best_best = axes[1][1]; next_next = axes[2][2]; last_last = axes[0][0];
				best_next = axes[1][2]; next_best = axes[2][1]; best_last = axes[1][0]; last_best = axes[0][1]; next_last = axes[2][0]; last_next = axes[0][2];
			}
			else
			{
				best_best = axes[2][2]; next_next = axes[0][0]; last_last = axes[1][1];
				best_next = axes[2][0]; next_best = axes[0][2]; best_last = axes[2][1]; last_best = axes[1][2]; next_last = axes[0][1]; last_next = axes[1][0];
			}
			const float pseudo_trace = ((1.0f + best_best) - next_next) - last_last;
			const float inv_pseudo_trace = 1.0f / sqrtf(pseudo_trace);
			const float half_inv_pseudo_trace = inv_pseudo_trace * 0.5f;
			const float q_best = (1.0f / inv_pseudo_trace) * 0.5f;
			const float q_next = half_inv_pseudo_trace * (best_next + next_best);
			const float q_last = half_inv_pseudo_trace * (best_last + last_best);
			q[3] = half_inv_pseudo_trace * (next_last - last_next);
			q[0] = best == 0 ? q_best : (best == 1 ? q_last : q_next);
			q[1] = best == 0 ? q_next : (best == 1 ? q_best : q_last);
			q[2] = best == 0 ? q_last : (best == 1 ? q_next : q_best);
		}

		qvv result;
		result.rotation = quat_normalize(make_float4(q[0], q[1], q[2], q[3]));
		result.translation = make_float4(translation[0], translation[1], translation[2], 0.0f);
		result.scale = make_float4(scale[0], scale[1], scale[2], 0.0f);
		return result;
	}

	// lhs first, then rhs (child, then parent); the quaternion path: callers route qvv_mul_takes_matrix_path() transforms through
	// qvv_mul_through_matrices
	__device__ __forceinline__ qvv qvv_mul(const qvv& lhs, const qvv& rhs)
	{
		qvv result;
		result.rotation = quat_mul(lhs.rotation, rhs.rotation);
		const float4 scaled = make_float4(lhs.translation.x * rhs.scale.x, lhs.translation.y * rhs.scale.y, lhs.translation.z * rhs.scale.z, 0.0f);
		const float4 rotated = quat_mul_vector3(scaled, rhs.rotation);
		result.translation = make_float4(rotated.x + rhs.translation.x, rotated.y + rhs.translation.y, rotated.z + rhs.translation.z, 0.0f);
		result.scale = make_float4(lhs.scale.x * rhs.scale.x, lhs.scale.y * rhs.scale.y, lhs.scale.z * rhs.scale.z, 0.0f);
		return result;
	}

	// the quaternion path of qvv_mul in ACLHIP_CONSUMERS_FAST arithmetic
	__device__ __forceinline__ qvv qvv_mul_fast(const qvv& lhs, const qvv& rhs)
	{
		qvv result;
		result.rotation = quat_mul_fast(lhs.rotation, rhs.rotation);
		const float4 scaled = make_float4(lhs.translation.x * rhs.scale.x, lhs.translation.y * rhs.scale.y, lhs.translation.z * rhs.scale.z, 0.0f);
		const float4 rotated = quat_mul_vector3_fast(scaled, rhs.rotation);
		result.translation = make_float4(rotated.x + rhs.translation.x, rotated.y + rhs.translation.y, rotated.z + rhs.translation.z, 0.0f);
		result.scale = make_float4(lhs.scale.x * rhs.scale.x, lhs.scale.y * rhs.scale.y, lhs.scale.z * rhs.scale.z, 0.0f);
		return result;
	}

	// apply_additive_to_base (core/additive_utils.h:150-160); format: acl::additive_clip_format8.
	// kMirrored = false compiles the matrix route of qvv_mul out (launches whose clips cannot produce a negative scale: its registers
	// would cost every launch a wave per SIMD).
	template<bool kMirrored>
	__device__ __forceinline__ qvv apply_additive_to_base(uint32_t additive_format, const qvv& base, const qvv& additive)
	{
		if (additive_format == 1)
		{
			qvv result = qvv_mul(additive, base);
			if (kMirrored && qvv_mul_takes_matrix_path(additive, base))
				result = qvv_mul_through_matrices(additive, base);
			return result;
		}
		if (additive_format == 2 || additive_format == 3)
		{
			// transform_add0 / transform_add1 (:128-142)
			qvv result;
			result.rotation = quat_mul(additive.rotation, base.rotation);
			result.translation = make_float4(additive.translation.x + base.translation.x, additive.translation.y + base.translation.y, additive.translation.z + base.translation.z, 0.0f);
			if (additive_format == 2)
				result.scale = make_float4(additive.scale.x * base.scale.x, additive.scale.y * base.scale.y, additive.scale.z * base.scale.z, 0.0f);
			else
				result.scale = make_float4((1.0f + additive.scale.x) * base.scale.x, (1.0f + additive.scale.y) * base.scale.y, (1.0f + additive.scale.z) * base.scale.z, 0.0f);
			return result;
		}
		return additive;
	}
}
