// host_experiments.inl -- part of aclhip.hip, compiled only with -DACLHIP_EXPERIMENTS: launches of the kernel variants of kernels_experiments.inl,
// each behind an environment knob (tools/exp_r3*.sh, tools/pmc_variants.sh).
//   ACLHIP_HANDOFF_DECODERS = n [ACLHIP_HANDOFF_LAST_ARRIVER=1]   decode waves + a store wave / the last arriver stores (one-shot grid)
//   ACLHIP_PERSISTENT = 31 / 71 / 62 / 142 / 151                  looping decode waves + store waves, <decoders><storers> per workgroup
//   ACLHIP_STAGED = 4 / 8                                         the staged kernel, waves per workgroup
//   ACLHIP_ITEMS_PER_WAVE = k [ACLHIP_ITEMS_PER_WAVE_WIDE=1]      k work items per wave in turn
// Each applies to poses of several windows; with the matching ..._ALWAYS=1 also to one-window poses.

namespace
{
	aclhip_status launch_experimental_tracks(aclhip_context* context, const aclhip_clip* clips, const float* sample_times, uint32_t num_instances,
		const decode_params& params, void* poses, uint64_t pose_stride_bytes, hipStream_t stream, uint32_t windows_per_instance, uint64_t num_waves,
		uint32_t num_blocks, bool any_settings, bool compact, uint32_t lds_quads_per_wave, size_t lds_bytes, bool& out_launched)
	{
		out_launched = true;
		// poses of several windows, common case, QVV48: decode waves hand their finished windows to one store wave per workgroup
		// (measurement aid: ACLHIP_HANDOFF_DECODERS = decode waves per workgroup, 0 = off; ACLHIP_HANDOFF_ALWAYS=1 also for one-window poses)
		static const uint32_t handoff_decoders = []() { const char* value = std::getenv("ACLHIP_HANDOFF_DECODERS"); return value != nullptr ? uint32_t(std::atol(value)) : k_handoff_default_decoders; }();
		static const bool handoff_always = []() { const char* value = std::getenv("ACLHIP_HANDOFF_ALWAYS"); return value != nullptr && value[0] == '1'; }();
		static const bool handoff_last_arriver = []() { const char* value = std::getenv("ACLHIP_HANDOFF_LAST_ARRIVER"); return value != nullptr && value[0] == '1'; }();
		if (handoff_decoders != 0 && !any_settings && !compact && (windows_per_instance > 1 || handoff_always))
		{
			void (*handoff_kernel)(ACLHIP_POSE_KERNEL_ARGUMENTS) = nullptr;
			if (!handoff_last_arriver)
				switch (handoff_decoders)
				{
				case 3: handoff_kernel = decompress_tracks_handoff_kernel<3>; break;
				case 4: handoff_kernel = decompress_tracks_handoff_kernel<4>; break;
				case 6: handoff_kernel = decompress_tracks_handoff_kernel<6>; break;
				case 7: handoff_kernel = decompress_tracks_handoff_kernel<7>; break;
				case 8: handoff_kernel = decompress_tracks_handoff_kernel<8>; break;
				case 15: handoff_kernel = decompress_tracks_handoff_kernel<15>; break;
				default: break;
				}
			else
				switch (handoff_decoders)
				{
				case 2: handoff_kernel = decompress_tracks_last_arriver_kernel<2>; break;
				case 3: handoff_kernel = decompress_tracks_last_arriver_kernel<3>; break;
				case 4: handoff_kernel = decompress_tracks_last_arriver_kernel<4>; break;
				case 6: handoff_kernel = decompress_tracks_last_arriver_kernel<6>; break;
				case 8: handoff_kernel = decompress_tracks_last_arriver_kernel<8>; break;
				default: break;
				}
			if (handoff_kernel == nullptr)
				return fail(context, ACLHIP_ERROR_INVALID_ARGUMENT, "ACLHIP_HANDOFF_DECODERS: no such kernel shape");
			const uint32_t handoff_blocks = uint32_t((num_waves + handoff_decoders - 1) / handoff_decoders);
			const size_t handoff_lds = 256 + size_t(lds_quads_per_wave) * 16 * handoff_decoders + (lds_bytes - size_t(lds_quads_per_wave) * 16 * k_waves_per_block);
			if (handoff_lds > 48 * 1024)
				ACLHIP_CHECK_HIP(context, hipFuncSetAttribute(reinterpret_cast<const void*>(handoff_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, int(handoff_lds)));
			hipLaunchKernelGGL(handoff_kernel, dim3(handoff_blocks), dim3((handoff_decoders + (handoff_last_arriver ? 0 : 1)) * k_wave_size), handoff_lds, stream,
				context->d_clips, context->d_clips_capacity, clips, sample_times, num_instances, windows_per_instance, params,
				static_cast<uint8_t*>(poses), pose_stride_bytes, lds_quads_per_wave, context->d_rejected);
			ACLHIP_CHECK_HIP(context, hipGetLastError());
			return ACLHIP_OK;
		}
		// persistent decode waves + store waves (kernels_pose.inl). Measurement knobs: ACLHIP_PERSISTENT = <decoders><storers> shape
		// (31 / 71 / 62 / 142 / 151), ACLHIP_PERSISTENT_ALWAYS=1 also for one-window poses
		static const uint32_t persistent_shape = []() { const char* value = std::getenv("ACLHIP_PERSISTENT"); return value != nullptr ? uint32_t(std::atol(value)) : k_persistent_default_shape; }();
		static const bool persistent_always = []() { const char* value = std::getenv("ACLHIP_PERSISTENT_ALWAYS"); return value != nullptr && value[0] == '1'; }();
		if (persistent_shape != 0 && !any_settings && !compact && (windows_per_instance > 1 || persistent_always))
		{
			void (*persistent_kernel)(ACLHIP_POSE_KERNEL_ARGUMENTS) = nullptr;
			uint32_t decoders = 0, storers = 0, slots = 0;
			switch (persistent_shape)
			{
			case 31: persistent_kernel = decompress_tracks_persistent_kernel<3, 1, 4>; decoders = 3; storers = 1; slots = 4; break;
			case 71: persistent_kernel = decompress_tracks_persistent_kernel<7, 1, 8>; decoders = 7; storers = 1; slots = 8; break;
			case 62: persistent_kernel = decompress_tracks_persistent_kernel<6, 2, 8>; decoders = 6; storers = 2; slots = 8; break;
			case 142: persistent_kernel = decompress_tracks_persistent_kernel<14, 2, 16>; decoders = 14; storers = 2; slots = 16; break;
			case 151: persistent_kernel = decompress_tracks_persistent_kernel<15, 1, 16>; decoders = 15; storers = 1; slots = 16; break;
			default: return fail(context, ACLHIP_ERROR_INVALID_ARGUMENT, "ACLHIP_PERSISTENT: no such kernel shape");
			}
			const size_t persistent_lds = sizeof(persistent_control) + size_t(lds_quads_per_wave) * 16 * slots;
			// as many workgroups as stay resident together: 32 wave slots and 160 KiB of LDS per CU
			const uint32_t per_cu = std::max<uint32_t>(std::min<uint32_t>(32u / (decoders + storers), uint32_t((160u * 1024u) / persistent_lds)), 1u);
			const uint32_t resident = context->num_compute_units * per_cu;
			const uint32_t persistent_blocks = uint32_t(std::min<uint64_t>((num_waves + decoders - 1) / decoders, resident));
			if (persistent_lds > 48 * 1024)
				ACLHIP_CHECK_HIP(context, hipFuncSetAttribute(reinterpret_cast<const void*>(persistent_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, int(persistent_lds)));
			hipLaunchKernelGGL(persistent_kernel, dim3(persistent_blocks), dim3((decoders + storers) * k_wave_size), persistent_lds, stream,
				context->d_clips, context->d_clips_capacity, clips, sample_times, num_instances, windows_per_instance, params,
				static_cast<uint8_t*>(poses), pose_stride_bytes, lds_quads_per_wave, context->d_rejected);
			ACLHIP_CHECK_HIP(context, hipGetLastError());
			return ACLHIP_OK;
		}
		// the staged kernel (kernels_pose.inl): keyframe bits staged through LDS, one base pose image per workgroup of consecutive instances.
		// ACLHIP_STAGED = waves per workgroup (4 / 8; 0 = off), ACLHIP_STAGED_ALWAYS=1 also for one-window poses
		static const uint32_t staged_waves = []() { const char* value = std::getenv("ACLHIP_STAGED"); return value != nullptr ? uint32_t(std::atol(value)) : k_default_staged_waves; }();
		static const bool staged_always = []() { const char* value = std::getenv("ACLHIP_STAGED_ALWAYS"); return value != nullptr && value[0] == '1'; }();
		if (staged_waves != 0 && !any_settings && !compact && (windows_per_instance > 1 || staged_always))
		{
			void (*staged_kernel)(ACLHIP_STAGED_KERNEL_ARGUMENTS) = staged_waves == 8 ? decompress_tracks_staged_kernel<8> : decompress_tracks_staged_kernel<4>;
			const uint32_t waves = staged_waves == 8 ? 8u : 4u;
			const uint32_t decoded_quads = std::max<uint32_t>(context->max_window_animated, 1);
			const uint32_t key_bytes = std::max<uint32_t>(context->max_window_key_bytes, 16);
			const size_t staged_lds = size_t(lds_quads_per_wave) * 16 + size_t(waves) * (size_t(decoded_quads) * 16 + size_t(key_bytes) * 2);
			if (staged_lds <= 64 * 1024)
			{
				const uint64_t groups = (uint64_t(num_instances) + waves - 1) / waves;
				if (groups * windows_per_instance > 0x7FFFFFFFull)
					return fail(context, ACLHIP_ERROR_INVALID_ARGUMENT, "batch too large: %u instances x %u pose windows", num_instances, windows_per_instance);
				if (staged_lds > 48 * 1024)
					ACLHIP_CHECK_HIP(context, hipFuncSetAttribute(reinterpret_cast<const void*>(staged_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, int(staged_lds)));
				hipLaunchKernelGGL(staged_kernel, dim3(uint32_t(groups * windows_per_instance)), dim3(waves * k_wave_size), staged_lds, stream,
					context->d_clips, context->d_clips_capacity, clips, sample_times, num_instances, windows_per_instance, params,
					static_cast<uint8_t*>(poses), pose_stride_bytes, lds_quads_per_wave, context->d_rejected, decoded_quads, key_bytes);
				ACLHIP_CHECK_HIP(context, hipGetLastError());
				return ACLHIP_OK;
			}
		}
		// several work items per wave, in turn (kernels_pose.inl: decompress_tracks_in_turn_r3_kernel). Measurement knobs:
		// ACLHIP_ITEMS_PER_WAVE = K (0 / 1 = one-shot), ACLHIP_ITEMS_PER_WAVE_ALWAYS=1 also for one-window poses
		static const uint32_t items_per_wave = []() { const char* value = std::getenv("ACLHIP_ITEMS_PER_WAVE"); return value != nullptr ? uint32_t(std::atol(value)) : k_default_items_per_wave; }();
		static const bool items_per_wave_always = []() { const char* value = std::getenv("ACLHIP_ITEMS_PER_WAVE_ALWAYS"); return value != nullptr && value[0] == '1'; }();
		if (items_per_wave > 1 && !any_settings && !compact && (windows_per_instance > 1 || items_per_wave_always))
		{
			decode_params turn_params = params;
			turn_params.items_per_wave = uint8_t(std::min<uint32_t>(items_per_wave, 255));
			// a wave's items are gridDim * 4 work items apart: a multiple of the windows per instance keeps its window index (and with it the
			// base pose window its LDS image holds) from turn to turn
			uint32_t turn_blocks = (num_blocks + turn_params.items_per_wave - 1) / turn_params.items_per_wave;
			while ((uint64_t(turn_blocks) * k_waves_per_block) % windows_per_instance != 0)
				turn_blocks++;
			// ACLHIP_ITEMS_PER_WAVE_WIDE = 1: 16 byte key reads at 8 waves per SIMD; 2 / 3: the same kernel with the registers of 7 / 6 waves per SIMD
			static const int turn_wide = []() { const char* value = std::getenv("ACLHIP_ITEMS_PER_WAVE_WIDE"); return value != nullptr ? int(value[0] - '0') : 0; }();
			hipLaunchKernelGGL(turn_wide == 1 ? decompress_tracks_in_turn_wide_loads_kernel : (turn_wide == 2 ? decompress_tracks_in_turn_wide_loads_7_kernel : (turn_wide == 3 ? decompress_tracks_in_turn_wide_loads_6_kernel : decompress_tracks_in_turn_r3_kernel)), dim3(turn_blocks), dim3(k_block_size), lds_bytes, stream,
				context->d_clips, context->d_clips_capacity, clips, sample_times, num_instances, windows_per_instance, turn_params,
				static_cast<uint8_t*>(poses), pose_stride_bytes, lds_quads_per_wave, context->d_rejected);
			ACLHIP_CHECK_HIP(context, hipGetLastError());
			return ACLHIP_OK;
		}
		out_launched = false;
		return ACLHIP_OK;
	}
}

// wall clock stamps of the last order_instances_grid_kernel launch: [workgroup][8] (tools/order_phases.py)
extern "C" int aclhip_exp_read_order_stamps(unsigned long long* out)
{
	return int(hipMemcpyFromSymbol(out, HIP_SYMBOL(aclhip::g_order_stamps), sizeof(unsigned long long) * 64 * 8));
}
