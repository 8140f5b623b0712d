// device_experiments.h -- part of aclhip_device.h under -DACLHIP_EXPERIMENTS (tools/build_experiments.sh): round 3's staged unpack,
// kept with the other measured-and-dropped variants (profiles/r03_experiments.md). Included inside namespace aclhip.
	// ---- the same unpack from keyframe bits STAGED IN LDS (the staged kernel, kernels_pose.inl) --------------------------------------------
	// A wave copies the runs of keyframe bits its window needs into LDS with coalesced 16 byte reads (window_span_entry); a lane then
	// finds its sub-track at bit `position` of its key's staging buffer, MSB first like the bitstream, and reads the same two windows
	// as unpack_animated_samples does from memory -- 8 bytes from the byte that holds the first bit for x and y, 4 bytes for z -- out of
	// LDS (gfx950 serves ds_read_b64 / b32 at any byte address), followed by the same instructions.
	typedef __attribute__((address_space(3))) const uint8_t* lds_bytes;

	__device__ __forceinline__ uint32_t load_be32(lds_bytes p)
	{
		uint32_t v;
		__builtin_memcpy(&v, (const __attribute__((address_space(3))) void*)p, 4);
		return __builtin_bswap32(v);
	}

	__device__ __forceinline__ uint64_t load_u64(lds_bytes p)
	{
		uint64_t v;
		__builtin_memcpy(&v, (const __attribute__((address_space(3))) void*)p, 8);
		return v;
	}

	// position0 / position1: bit position of the sub-track's first bit inside key_bytes0 / key_bytes1 (anything valid for a width of 0)
	template<bool kHasRaw>
	__device__ __forceinline__ void unpack_staged_samples(lds_bytes key_bytes0, lds_bytes key_bytes1, uint32_t position0, uint32_t position1,
		const plan_entry& plan0, const plan_entry& plan1, const clip_range_entry& clip_range, bool is_rotation, float out_v0[3], float out_v1[3])
	{
		const uint32_t num_bits0 = plan0.bit_offset_and_width >> 24;
		const uint32_t num_bits1 = plan1.bit_offset_and_width >> 24;
		const uint32_t position_z0 = position0 + 2u * num_bits0;
		const uint32_t position_z1 = position1 + 2u * num_bits1;

		const uint64_t window_xy0 = load_u64(key_bytes0 + (position0 >> 3));
		const uint32_t window_z0 = load_be32(key_bytes0 + (position_z0 >> 3));
		const uint64_t window_xy1 = load_u64(key_bytes1 + (position1 >> 3));
		const uint32_t window_z1 = load_be32(key_bytes1 + (position_z1 >> 3));

		float v[2][3];
		#pragma unroll
		for (uint32_t key = 0; key < 2; ++key)
		{
			const uint64_t window_xy = key == 0 ? window_xy0 : window_xy1;
			const uint32_t hi_z = key == 0 ? window_z0 : window_z1;
			const uint32_t num_bits = key == 0 ? num_bits0 : num_bits1;
			const uint32_t shift_xy = (key == 0 ? position0 : position1) & 7u;
			const uint32_t shift_z = (key == 0 ? position_z0 : position_z1) & 7u;
			const plan_entry& plan = key == 0 ? plan0 : plan1;

			const uint32_t hi = __builtin_bswap32(uint32_t(window_xy));
			const uint32_t lo = __builtin_bswap32(uint32_t(window_xy >> 32));
			const uint32_t x = __builtin_amdgcn_ubfe(hi, 32u - shift_xy - num_bits, num_bits);
			const uint32_t window_y = __builtin_amdgcn_alignbit(hi, lo, 32u - (shift_xy + num_bits));
			const uint32_t y = __builtin_amdgcn_ubfe(window_y, 32u - num_bits, num_bits);
			const uint32_t z = __builtin_amdgcn_ubfe(hi_z, 32u - shift_z - num_bits, num_bits);

			const float quantized[3] = { float(x) * plan.inv_max_value, float(y) * plan.inv_max_value, float(z) * plan.inv_max_value };
			#pragma unroll
			for (uint32_t c = 0; c < 3; ++c)
			{
				const float segment_value = (quantized[c] * plan.range_extent[c]) + plan.range_min[c];
				v[key][c] = (segment_value * clip_range.range_extent[c]) + clip_range.range_min[c];
			}
		}

		if (kHasRaw)
		{
			// raw fp32 keyframes (unpack_animated_samples): three big endian floats from an arbitrary bit on; no range expansion
			#pragma unroll
			for (uint32_t key = 0; key < 2; ++key)
			{
				if ((key == 0 ? num_bits0 : num_bits1) == 32u)
				{
					const uint32_t position = key == 0 ? position0 : position1;
					lds_bytes bytes = (key == 0 ? key_bytes0 : key_bytes1) + (position >> 3);
					const uint32_t shift = position & 7u;
					const uint32_t w0 = load_be32(bytes), w1 = load_be32(bytes + 4), w2 = load_be32(bytes + 8), w3 = load_be32(bytes + 12);
					float raw[3] = { __uint_as_float(__funnelshift_l(w1, w0, shift)), __uint_as_float(__funnelshift_l(w2, w1, shift)), __uint_as_float(__funnelshift_l(w3, w2, shift)) };
					#pragma unroll
					for (uint32_t c = 0; c < 3; ++c)
						v[key][c] = is_rotation ? (((raw[c] * 1.0f) + 0.0f) * 1.0f) + 0.0f : raw[c];
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
