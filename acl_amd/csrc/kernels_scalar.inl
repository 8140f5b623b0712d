// kernels_scalar.inl -- part of aclhip.hip (one translation unit; included there, in this order, not compiled on its own).
// Scalar track lists: decompress_scalar_tracks_kernel, decompress_scalar_track_kernel.

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
	__device__ __forceinline__ void decode_scalar_track_values(const scalar_frames& frames, const scalar_track_tables<C>& tables, float alpha, float* value)
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
	}

	template<uint32_t C, bool kFromLds>
	__device__ __forceinline__ void decode_scalar_track(const scalar_frames& frames, const scalar_track_tables<C>& tables, float alpha, float* destination)
	{
		float value[C];
		decode_scalar_track_values<C, kFromLds>(frames, tables, alpha, value);
		store_streaming_floats<C>(destination, value);
	}

	// Which tracks a lane of a track list wave takes: kRows of them. Rows of one or two floats per track are written 16 bytes per lane
	// -- 4 / C CONSECUTIVE tracks per lane, one dwordx4 store (1 KiB per store instruction instead of 256 / 512 bytes) -- when the wave
	// takes 4 tracks per lane; otherwise (and for 3 / 4 floats per track, whose rows already are 12 / 16 bytes per lane) 64 apart.
	template<uint32_t C, uint32_t kRows>
	struct scalar_lane_tracks
	{
		static constexpr uint32_t k_run = (C <= 2 && kRows * C >= 4) ? 4 / C : 1;		// consecutive tracks per lane
		static __device__ __forceinline__ uint32_t track_of(uint32_t first_track, uint32_t j, uint32_t lane)
		{
			return first_track + (j / k_run) * (k_wave_size * k_run) + lane * k_run + (j % k_run);
		}
	};

	// Decodes a lane's tracks of ONE instance and stores them (row: wave uniform); rounding: per track policy source (null: the alpha as it is)
	template<uint32_t C, uint32_t kRows, bool kFromLds, bool kPolicies>
	__device__ __forceinline__ void decode_and_store_lane_tracks(const scalar_frames& frames, const scalar_track_tables<C> (&tables)[kRows], float seek_alpha,
		uint32_t rounding_policy, const uint8_t* track_rounding_policies, uint32_t first_track, uint32_t num_tracks, uint32_t lane, float* row)
	{
		typedef scalar_lane_tracks<C, kRows> lane_tracks;
		constexpr uint32_t k_run = lane_tracks::k_run;
		#pragma unroll
		for (uint32_t j0 = 0; j0 < kRows; j0 += k_run)
		{
			float values[k_run * C];
			#pragma unroll
			for (uint32_t r = 0; r < k_run; ++r)
			{
				const uint32_t track_index = lane_tracks::track_of(first_track, j0 + r, lane);
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
					decode_scalar_track_values<C, kFromLds>(frames, tables[j0 + r], alpha, values + r * C);
				}
			}
			const uint32_t first_of_run = lane_tracks::track_of(first_track, j0, lane);
			if constexpr (k_run == 1)
			{
				if (first_of_run < num_tracks)
				{
					float value[C];
					#pragma unroll
					for (uint32_t c = 0; c < C; ++c)
						value[c] = values[c];
					store_streaming_floats_at<C>(row, first_of_run * C * 4u, value);
				}
			}
			else
			{
				if (first_of_run + k_run <= num_tracks)
				{
					const float value[4] = { values[0], values[1], values[2], values[3] };
					store_streaming_floats_at<4>(row, first_of_run * C * 4u, value);		// (rows are 16 byte aligned, runs start at multiples of 4 floats)
				}
				else
				{
					#pragma unroll
					for (uint32_t r = 0; r < k_run; ++r)
						if (first_of_run + r < num_tracks)
						{
							float value[C];
							#pragma unroll
							for (uint32_t c = 0; c < C; ++c)
								value[c] = values[r * C + c];
							store_streaming_floats_at<C>(row, (first_of_run + r) * C * 4u, value);
						}
				}
			}
		}
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

	// The shape of a scalar launch (waves per instance, LDS per key frame) is decided when it is enqueued, from the batch's row stride
	// and the registry (launch_scalar); the list an instance names is read when the wave runs. A list with more tracks than the
	// launch has waves for, a frame larger than its LDS slot, a row wider than the caller's stride -- a list registered behind a
	// captured launch's back, a stride too small for a list of the batch -- is refused and counted (see launch_refuses_clip).
	template<bool kFromLds>
	__device__ __forceinline__ bool scalar_launch_refuses_clip(const device_clip& clip, uint32_t tracks_per_wave, uint32_t chunks_per_instance, uint32_t frame_lds_bytes, uint64_t out_stride_bytes)
	{
		const uint32_t num_components = (clip.flags >> k_clip_components_shift) & 7u;
		return uint64_t(chunks_per_instance) * tracks_per_wave < clip.num_tracks
			|| uint64_t(clip.num_tracks) * num_components * 4u > out_stride_bytes
			|| (kFromLds && ((clip.num_animated + 7u) >> 3) + 24u > frame_lds_bytes);
	}

	// frame_lds_bytes != 0: both key frames' bits are DMA'd into LDS (one coalesced global_load_lds per KiB) while the track tables
	// are fetched, and every bit field is two aligned LDS dwords away -- instead of 2 * C scattered, unaligned global reads per track
	// through the texture unit, dependent on the table read. frame_lds_bytes == 0 (a registered list's frame does not fit): global reads.
	// kRows: tracks per lane (1 when no registered list has more than 64 tracks, else 4); kPolicies: per track rounding -- launch wide
	// facts, compiled as separate kernels so that each stays small.
	// kComponents: 0 = lists of any track type; 1 = every registered list is float1f (launch wide, host_scalar_misc.inl): one code path
	template<bool kFromLds, uint32_t kRows, bool kPolicies, uint32_t kComponents = 0>
	__device__ __forceinline__ void decompress_scalar_tracks_instance(const device_clip* __restrict__ clips, uint32_t num_clips,
		const uint32_t* __restrict__ clip_ids, const float* __restrict__ sample_times, uint32_t instance, uint32_t chunk,
		const decode_params& params, uint8_t* __restrict__ out, uint64_t out_stride_bytes, uint32_t frame_lds_bytes, uint8_t* wave_lds, uint32_t lane,
		unsigned long long* __restrict__ rejected_count, uint32_t chunks_per_instance)
	{
		constexpr uint32_t k_tracks_per_wave = kRows * k_wave_size;

		const uint32_t clip_id = as_constant(clip_ids)[instance];
		const float sample_time = as_constant(sample_times)[instance];
		const device_clip clip = load_clip(clips, clip_id < num_clips ? clip_id : 0);
		if (clip_id >= num_clips || !is_scalar_clip(clip.flags) || scalar_launch_refuses_clip<kFromLds>(clip, k_tracks_per_wave, chunks_per_instance, frame_lds_bytes, out_stride_bytes))
		{
			if (lane == 0 && chunk == 0)
				atomicAdd(rejected_count, 1ull);
			return;
		}

		const uint32_t first_track = chunk * k_tracks_per_wave;
		if (first_track >= clip.num_tracks || clip.num_samples == 0)
			return;		// past the end of this clip's track list / empty track list (:185-186,246-248)

		const uint32_t rounding_policy = __builtin_amdgcn_readfirstlane(instance_rounding_policy_of(params, instance));
		const uint32_t looping_policy = __builtin_amdgcn_readfirstlane(instance_looping_policy_of(params, instance));

		// seek_v0 (:182-240): a frame is num_bits_per_frame bits
		uint32_t key_frame0, key_frame1;
		float seek_alpha;
		find_key_frames(clip.flags, clip.num_samples, clip.sample_rate, clip.duration_clamp, clip.duration_wrap, sample_time, rounding_policy, looping_policy,
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
			uint8_t* lds = wave_lds;
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
		const uint8_t* const track_rounding_policies = instance_track_rounding_of(params, instance);		// (wave uniform)

		// Every lane takes kRows tracks, 64 apart. The component count is wave uniform: one specialised loop runs.
		const auto decode_tracks = [=](auto components)
		{
			constexpr uint32_t C = decltype(components)::value;

			// the table rows travel together with the frame copies ...
			scalar_track_tables<C> tables[kRows];
			#pragma unroll
			for (uint32_t j = 0; j < kRows; ++j)
				tables[j] = load_scalar_track_tables<C>(headers, ranges, min(scalar_lane_tracks<C, kRows>::track_of(first_track, j, lane), num_tracks - 1));

			if (kFromLds)
			{
				// ... which must have landed before any lane reads a field
				__builtin_amdgcn_s_waitcnt(0);
				__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
				__builtin_amdgcn_wave_barrier();
				__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
			}

			decode_and_store_lane_tracks<C, kRows, kFromLds, kPolicies>(frames, tables, seek_alpha, rounding_policy, track_rounding_policies, first_track, num_tracks, lane, row);
		};
		if constexpr (kComponents == 1)
		{
			if (num_components == 1)
				decode_tracks(std::integral_constant<uint32_t, 1>());
			else if (lane == 0 && chunk == 0)
				atomicAdd(rejected_count, 1ull);		// (a wider list registered behind the launch's back)
		}
		else
		{
			switch (num_components)
			{
			case 1: decode_tracks(std::integral_constant<uint32_t, 1>()); break;
			case 2: decode_tracks(std::integral_constant<uint32_t, 2>()); break;
			case 3: decode_tracks(std::integral_constant<uint32_t, 3>()); break;
			default: decode_tracks(std::integral_constant<uint32_t, 4>()); break;
			}
		}
	}

	template<bool kFromLds, uint32_t kRows, bool kPolicies>
	__global__ __launch_bounds__(k_block_size) void decompress_scalar_tracks_kernel(const device_clip* __restrict__ clips, uint32_t num_clips,
		const uint32_t* __restrict__ clip_ids, const float* __restrict__ sample_times, uint32_t num_instances, uint32_t chunks_per_instance,
		decode_params params, uint8_t* __restrict__ out, uint64_t out_stride_bytes, uint32_t frame_lds_bytes, unsigned long long* __restrict__ rejected_count)
	{
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
		decompress_scalar_tracks_instance<kFromLds, kRows, kPolicies>(clips, num_clips, clip_ids, sample_times, instance, chunk, params, out, out_stride_bytes, frame_lds_bytes,
			dynamic_lds + size_t(wave_in_block) * 2u * frame_lds_bytes, lane, rejected_count, chunks_per_instance);
	}

	// A track list's output row is small (256 float1f curves: 1 KiB) next to what a wave reads to produce it -- 4 KiB of track tables
	// and a chain of three dependent memory round trips (instance -> clip record -> frames + tables). One wave therefore takes
	// k_scalar_group CONSECUTIVE instances: when they are of one clip (the usual shape of an instance list: sorted, or one list for
	// all characters) the tables are fetched once and stay in registers, the seeks run back to back on the scalar unit, all the key
	// frames are DMA'd into LDS together, and the wave pays its round trips once per group instead of once per instance
	// (28.0 -> 25.8 us for 64k x 256 float1f curves, 25.0 with 16 byte stores: what remains is the decode's own arithmetic, about 140
	// VALU instructions per instance and 256 curves = 17 us of issue per SIMD; groups of 2 / 8: 28.4 / 32.6 us; capping the registers
	// for 6 / 8 waves per SIMD: 22.7 / 42 us on this shape, slower on 1024 curves). Groups of mixed clips fall back to one instance after the other.
	// (Round 5 let a wave take several groups IN TURN, keeping a float1f launch's tables in registers from turn to turn, to break the
	// lock step of two rounds of waves: 2 / 4 turns at full residency 26.6 / 29.7 us against 25.3 us for this one-shot form on the same
	// box, groups of 2 or 1 instance worse still -- profiles/r05_experiments.md 5; the launch's own instruction count is 19.7 us of vector
	// issue at the PEAK clock under a 22 - 23 us launch.)
	constexpr uint32_t k_scalar_group = 4;

	template<uint32_t kRows, bool kPolicies, uint32_t kComponents>
	__global__ __launch_bounds__(k_block_size) void decompress_scalar_tracks_grouped_kernel(const device_clip* __restrict__ clips, uint32_t num_clips,
		const uint32_t* __restrict__ clip_ids, const float* __restrict__ sample_times, uint32_t num_instances, uint32_t chunks_per_instance,
		decode_params params, uint8_t* __restrict__ out, uint64_t out_stride_bytes, uint32_t frame_lds_bytes, unsigned long long* __restrict__ rejected_count)
	{
		constexpr uint32_t k_tracks_per_wave = kRows * k_wave_size;
		extern __shared__ __attribute__((aligned(16))) uint8_t dynamic_lds[];
		const uint32_t lane = threadIdx.x & (k_wave_size - 1);
		const uint32_t wave_in_block = __builtin_amdgcn_readfirstlane(threadIdx.x / k_wave_size);
		const uint32_t work_item = blockIdx.x * k_waves_per_block + wave_in_block;
		uint32_t group = work_item;
		uint32_t chunk = 0;
		if (chunks_per_instance != 1)
		{
			group = work_item / chunks_per_instance;
			chunk = work_item - group * chunks_per_instance;
		}
		const uint32_t first_instance = group * k_scalar_group;
		if (first_instance >= num_instances)
			return;
		const uint32_t count = min(k_scalar_group, num_instances - first_instance);
		uint8_t* wave_lds = dynamic_lds + size_t(wave_in_block) * 2u * k_scalar_group * frame_lds_bytes;

		// the group's clips: one and the same (valid scalar) clip?
		uint32_t ids[k_scalar_group];
		float times[k_scalar_group];
		#pragma unroll
		for (uint32_t k = 0; k < k_scalar_group; ++k)
		{
			ids[k] = as_constant(clip_ids)[first_instance + min(k, count - 1)];
			times[k] = as_constant(sample_times)[first_instance + min(k, count - 1)];
		}
		bool uniform = ids[0] < num_clips;
		#pragma unroll
		for (uint32_t k = 1; k < k_scalar_group; ++k)
			uniform = uniform && ids[k] == ids[0];
		const device_clip clip = load_clip(clips, uniform ? ids[0] : 0);
		const uint32_t first_track = chunk * k_tracks_per_wave;
		uniform = uniform && is_scalar_clip(clip.flags) && first_track < clip.num_tracks && clip.num_samples != 0
			&& !scalar_launch_refuses_clip<true>(clip, k_tracks_per_wave, chunks_per_instance, frame_lds_bytes, out_stride_bytes);
		if (!uniform)
		{
			// mixed, refused or empty: one instance after the other through the first frame slots (each call waits for its own LDS reads)
			for (uint32_t k = 0; k < count; ++k)
			{
				decompress_scalar_tracks_instance<true, kRows, kPolicies, kComponents>(clips, num_clips, clip_ids, sample_times, first_instance + k, chunk, params, out, out_stride_bytes,
					frame_lds_bytes, wave_lds, lane, rejected_count, chunks_per_instance);
				wave_lds_barrier();
			}
			return;
		}

		const uint32_t num_components = (clip.flags >> k_clip_components_shift) & 7u;
		const uint32_t num_bits_per_frame = clip.num_animated;
		const uint32_t animated_bit_base = clip.num_segments * 8u;		// scalar clips: byte offset of the animated values in the blob

		scalar_frames frames[k_scalar_group];
		float alphas[k_scalar_group];
		uint32_t rounding_policies[k_scalar_group];
		#pragma unroll
		for (uint32_t k = 0; k < k_scalar_group; ++k)
		{
			rounding_policies[k] = __builtin_amdgcn_readfirstlane(instance_rounding_policy_of(params, first_instance + min(k, count - 1)));
			const uint32_t looping_policy = __builtin_amdgcn_readfirstlane(instance_looping_policy_of(params, first_instance + min(k, count - 1)));
			uint32_t key_frame0, key_frame1;
			find_key_frames(clip.flags, clip.num_samples, clip.sample_rate, clip.duration_clamp, clip.duration_wrap, times[k], rounding_policies[k], looping_policy,
				key_frame0, key_frame1, alphas[k]);
			frames[k] = scalar_frames();
			frames[k].blob = clip.blob;
			frames[k].frame_bit_offset[0] = key_frame0 * num_bits_per_frame;
			frames[k].frame_bit_offset[1] = key_frame1 * num_bits_per_frame;
			// Two key frames that follow each other in the blob (the usual case: key_frame1 = key_frame0 + 1, or the same frame at the
			// ends of the clip) travel as ONE copy into the instance's two slots (half the copy instructions: 64 x 16 bytes each);
			// a pair that wraps around the end of a looping clip travels as two.
			const bool adjacent = key_frame1 - key_frame0 <= 1u;
			#pragma unroll
			for (uint32_t key = 0; key < 2; ++key)
			{
				const uint32_t first_bit = animated_bit_base + frames[k].frame_bit_offset[adjacent ? 0 : key];
				const uint32_t last_bit = animated_bit_base + frames[k].frame_bit_offset[adjacent ? 1 : key] + num_bits_per_frame;
				const uint32_t first_byte = (first_bit >> 3) & ~15u;
				const uint32_t num_bytes = (((last_bit + 7u) >> 3) + 8u) - first_byte;
				uint8_t* destination = wave_lds + (k * 2u + (adjacent ? 0u : key)) * frame_lds_bytes;
				if (k < count && (key == 0 || !adjacent))
					for (uint32_t base = 0; base < num_bytes; base += k_wave_size * 16u)
					{
						if (base + lane * 16u < num_bytes)
							__builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(clip.blob + first_byte + base + lane * 16u),
								(__attribute__((address_space(3))) void*)(destination + base), 16, 0, 0);
					}
				frames[k].lds_frame[key] = reinterpret_cast<const uint32_t*>(destination);
				frames[k].lds_bit_base[key] = first_byte * 8u;
			}
		}

		const scalar_track_header* const headers = reinterpret_cast<const scalar_track_header*>(clip.plan);
		const float* const ranges = reinterpret_cast<const float*>(clip.clip_ranges);
		const uint32_t num_tracks = clip.num_tracks;
		const auto decode_tracks = [&](auto components)
		{
			constexpr uint32_t C = decltype(components)::value;
			scalar_track_tables<C> tables[kRows];
			#pragma unroll
			for (uint32_t j = 0; j < kRows; ++j)
				tables[j] = load_scalar_track_tables<C>(headers, ranges, min(scalar_lane_tracks<C, kRows>::track_of(first_track, j, lane), num_tracks - 1));
			wave_lds_barrier();		// every frame copy has landed

			#pragma unroll
			for (uint32_t k = 0; k < k_scalar_group; ++k)
			{
				if (k >= count)
					break;
				float* row = reinterpret_cast<float*>(out + uint64_t(first_instance + k) * out_stride_bytes);
				decode_and_store_lane_tracks<C, kRows, true, kPolicies>(frames[k], tables, alphas[k], rounding_policies[k], kPolicies ? instance_track_rounding_of(params, first_instance + k) : nullptr,
					first_track, num_tracks, lane, row);
			}
		};
		if constexpr (kComponents == 1)
		{
			if (num_components == 1)
				decode_tracks(std::integral_constant<uint32_t, 1>());
			else if (lane == 0 && chunk == 0)
				atomicAdd(rejected_count, (unsigned long long)count);
		}
		else
		{
			switch (num_components)
			{
			case 1: decode_tracks(std::integral_constant<uint32_t, 1>()); break;
			case 2: decode_tracks(std::integral_constant<uint32_t, 2>()); break;
			case 3: decode_tracks(std::integral_constant<uint32_t, 3>()); break;
			default: decode_tracks(std::integral_constant<uint32_t, 4>()); break;
			}
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

		const uint32_t rounding_policy = instance_rounding_policy_of(params, instance);
		uint32_t key_frame0, key_frame1;
		float alpha;
		find_key_frames(flags, clip.num_samples, clip.sample_rate, clip.duration_clamp, clip.duration_wrap, sample_times[instance], rounding_policy, instance_looping_policy_of(params, instance),
			key_frame0, key_frame1, alpha);
		if (params.per_track_rounding != 0)
		{
			uint32_t policy = rounding_policy;
			if (rounding_policy == k_round_per_track)
			{
				const uint8_t* track_policies = instance_track_rounding_of(params, instance);
				policy = track_policies != nullptr ? track_policies[track_index] : k_round_none;
			}
			alpha = apply_rounding_policy(alpha, policy);
		}

		const uint32_t num_bits_per_frame = clip.num_animated;
		decode_scalar_track_any((flags >> k_clip_components_shift) & 7u, clip.blob, reinterpret_cast<const scalar_track_header*>(clip.plan),
			reinterpret_cast<const float*>(clip.clip_ranges), track_index, key_frame0 * num_bits_per_frame, key_frame1 * num_bits_per_frame, alpha,
			reinterpret_cast<float*>(out + uint64_t(instance) * out_stride_bytes));
	}
