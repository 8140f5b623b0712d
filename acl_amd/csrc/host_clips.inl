// host_clips.inl -- part of aclhip.hip (one translation unit; included there, in this order, not compiled on its own).
// Host side: clip registration (registration time tables), unregistration.

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

	std::lock_guard<std::shared_mutex> lock(context->mutex);
	device_guard guard(context->device);
	if (!guard.ok)
		return fail(context, ACLHIP_ERROR_DEVICE, "hipSetDevice(%d) failed", context->device);
	collect_retired(context, false);

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

	size_t staging_used = 0;
	if (!stage_upload(context, d_memory, staging.data(), total_bytes, staging_used)
		|| !stage_upload(context, context->d_clips + slot, &record, sizeof(record), staging_used) || !finish_uploads(context))
	{
		free_clip_memory(context, d_memory);
		context->free_slots.push_back(slot);
		return fail(context, ACLHIP_ERROR_DEVICE, "uploading the clip failed");
	}
	context->clips_registered++;

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
	entry.scalar_tracks = num_tracks;
	entry.scalar_frame_bytes = (num_bits_per_frame + 7) / 8;
	context->max_scalar_tracks = std::max(context->max_scalar_tracks, entry.scalar_tracks);
	context->max_scalar_frame_bytes = std::max(context->max_scalar_frame_bytes, entry.scalar_frame_bytes);
	entry.wide_scalar = num_components != 1;
	context->num_wide_scalar_clips += entry.wide_scalar ? 1u : 0u;

	*out_clip = slot;
	return ACLHIP_OK;
}

namespace
{
	// The device copy of a database's list of bound clips (refresh_database_sample_tiers_kernel): rewritten whole, on the copy stream,
	// the calling thread waits. A list that outgrows its buffer moves to a new one; the old one is retired behind the work in flight.
	aclhip_status upload_bound_clips(aclhip_context* context, host_database& db)
	{
		if (db.bound_clips.size() > db.bound_clips_capacity)
		{
			const uint32_t capacity = std::max<uint32_t>(uint32_t(db.bound_clips.size()) * 2, 64);
			uint32_t* grown = nullptr;
			if (hipMalloc(reinterpret_cast<void**>(&grown), size_t(capacity) * sizeof(uint32_t)) != hipSuccess)
				return fail(context, ACLHIP_ERROR_OUT_OF_MEMORY, "hipMalloc of a database's clip list failed");
			if (db.d_bound_clips != nullptr)
			{
				aclhip_context::retired_item item;
				item.device_memory = db.d_bound_clips;
				retire(context, std::move(item));
			}
			db.d_bound_clips = grown;
			db.bound_clips_capacity = capacity;
		}
		if (db.bound_clips.empty())
			return ACLHIP_OK;
		size_t staging_used = 0;
		if (!stage_upload(context, db.d_bound_clips, db.bound_clips.data(), db.bound_clips.size() * sizeof(uint32_t), staging_used) || !finish_uploads(context))
			return fail(context, ACLHIP_ERROR_DEVICE, "uploading a database's clip list failed");
		return ACLHIP_OK;
	}

	// ---- What the VALUES of a clip's rotations allow the kernels: may they use the SHORT correctly rounded square root / reciprocal? ----
	// sqrt_rn_short / rcp_rn_short (aclhip_device.h) give the bits of sqrtf / 1.0f / x on x == 0 or x >= 2^-96, and on 2^-126 <= x <=
	// 2^126 (checked on every float, tools/probes/exact_math_probe.hip); the compiler's general forms cost 16 / 11 instructions instead
	// of 9 / 5 because they also cover what lies outside. Whether a clip can ever hand the kernels such an argument is decided HERE,
	// once. With a = fl(1 - x^2), b = fl(a - y^2), c = fl(b - z^2), W^2 = |c|: a nonzero difference of two floats is a multiple of the
	// smaller one's ulp, so a is 0 or >= 2^-24; b can only be small when a and y^2 nearly cancel (both >= 2^-24: |b| = 0 or >= 2^-48) or
	// when a == 0 (b = -y^2); c likewise is 0, a multiple of an ulp >= 2^-72, or the plain sum -(|b| + z^2). So |c| lands in (0, 2^-96)
	// only as the square of a component of magnitude below 2^-48 behind an EXACT cancellation in front of it -- a clip none of whose
	// animated rotation components can decode to a NONZERO value below 2^-47 in magnitude, and whose values are bounded, is safe; the norms the
	// normalize and the object space walk meet are then within [0.49, 2^41]. A component's decoded value is a monotone function of its
	// quantized field (rounding is monotone, the extents are checked non negative), so the two field values next to its zero
	// crossing decide: two binary searches per (segment, rotation, component). Ranges that are negative, not finite or huge and
	// constant rotations beyond 2^20 make the clip "not provably safe": its waves take the compiler's forms. Samples stored RAW (any
	// float) are not judged here: a wave that meets one takes the compiler's forms anyway (kHasRaw), and k_clip_raw_rotations tells the
	// object space walk that this clip's rotations are only as good as their normalization. Either way the poses are bit identical
	// to the reference's.
	struct rotation_value_facts
	{
		bool short_exact_math;		// -> k_clip_short_exact_math
		bool raw_rotations;			// -> k_clip_raw_rotations
	};
	rotation_value_facts analyze_rotation_values(const uint8_t* blob, uint32_t blob_size, bool key_frames_in_a_database, bool key_frames_stripped, uint32_t num_tracks, uint32_t num_samples, uint32_t num_segments,
		uint32_t num_animated, const std::vector<plan_entry>& plan, const std::vector<clip_range_entry>& clip_ranges, const std::vector<sample_record>& samples, const std::vector<float>& base_pose)
	{
		bool short_exact_math = true, raw_rotations = false, grid_has_tiny_values = false;
		if (num_tracks != 0)
		{
			constexpr float k_tiny = 7.1054273576010019e-15f;		// 2^-47 (a binade above what the argument needs)
			constexpr float k_huge = 1048576.0f;					// 2^20
			const auto decoded = [](const plan_entry& entry, const clip_range_entry& range, uint32_t c, uint32_t field)
			{
				// the kernels' own operations, in their order (unpack_animated_samples, aclhip_device.h)
				const float quantized = float(field) * entry.inv_max_value;
				const float segment_value = (quantized * entry.range_extent[c]) + entry.range_min[c];
				return (segment_value * range.range_extent[c]) + range.range_min[c];
			};
			for (uint32_t a = 0; a < num_animated && short_exact_math; ++a)
			{
				const clip_range_entry& range = clip_ranges[a];
				if (range.quad_index != range.track_index * 3)
					continue;
				for (uint32_t si = 0; si < num_segments && short_exact_math; ++si)
				{
					const plan_entry& entry = plan[size_t(si) * num_animated + a];
					const uint32_t num_bits = entry.bit_offset_and_width >> 24;
					if (is_raw_width(num_bits))
					{
						// any floats: a wave that meets a raw sample takes the compiler's forms (decode_animated_sub_track<kHasRaw = true>);
						// the rest of the clip is judged on its quantized samples
						raw_rotations = true;
						continue;
					}
					const uint32_t max_field = num_bits == 0 ? 0u : (1u << num_bits) - 1u;
					for (uint32_t c = 0; c < 3 && short_exact_math; ++c)
					{
						const bool ordered = entry.range_extent[c] >= 0.0f && range.range_extent[c] >= 0.0f && std::isfinite(entry.range_min[c]) && std::isfinite(range.range_min[c]);
						const float lowest = decoded(entry, range, c, 0), highest = decoded(entry, range, c, max_field);
						if (!ordered || !(std::fabs(lowest) <= k_huge) || !(std::fabs(highest) <= k_huge))
						{
							short_exact_math = false;
							break;
						}
						// the smallest field that decodes to a value > 0 / >= 0 (max_field + 1: none)
						const auto first_field = [&](bool strictly)
						{
							uint32_t low = 0, high = max_field + 1;
							while (low < high)
							{
								const uint32_t middle = low + (high - low) / 2;
								const float value = decoded(entry, range, c, middle);
								if (strictly ? value > 0.0f : value >= 0.0f)
									high = middle;
								else
									low = middle + 1;
							}
							return low;
						};
						const uint32_t first_positive = first_field(true), first_non_negative = first_field(false);
						if (first_positive <= max_field && decoded(entry, range, c, first_positive) < k_tiny)
							grid_has_tiny_values = true;
						if (first_non_negative > 0 && decoded(entry, range, c, first_non_negative - 1) > -k_tiny)
							grid_has_tiny_values = true;
					}
				}
			}
			// The grid holds a value within 2^-47 of zero (the idle axes of a hinge joint: +-1e-7 of noise in steps of 1e-11): that no SAMPLE
			// need ever take. The exact criterion then, on the key frames the clip actually stores: W^2 of every one of them, computed the way
			// the kernels do, is 0 or >= 2^-96. (A millisecond for a 300-bone clip, paid only by clips the grid test refuses. Key frames that
			// live in a database are not in this buffer: such clips stay refused.)
			if (short_exact_math && grid_has_tiny_values)
			{
				short_exact_math = !key_frames_in_a_database;
				constexpr float k_gap = 1.2621774483536189e-29f;		// 2^-96
				const auto read_field = [&](uint64_t bit, uint32_t num_bits) -> uint32_t
				{
					// num_bits <= 23 big endian bits at any bit address of the blob (bytes past its end read as zero)
					uint64_t window = 0;
					for (uint32_t i = 0; i < 5; ++i)
					{
						const uint64_t byte = (bit >> 3) + i;
						window = (window << 8) | (byte < blob_size ? blob[byte] : 0u);
					}
					return uint32_t((window >> (40u - (bit & 7u) - num_bits)) & ((uint64_t(1) << num_bits) - 1u));
				};
				for (uint32_t sample = 0; sample < num_samples && short_exact_math; ++sample)
				{
					const sample_record& record = samples[sample];
					const uint32_t si = record.segment_and_local >> 5, local = record.segment_and_local & 31u;
					// (stripped key frames: bit `31 - local` of sample_indices says whether this one is stored, the ones in front of it where;
					// nothing stripped: sample_indices IS the sample's index inside its segment)
					if (key_frames_stripped && (record.sample_indices & (0x80000000u >> local)) == 0)
						continue;
					const uint32_t stored_ordinal = key_frames_stripped ? uint32_t(__builtin_popcount(record.sample_indices & ~(0xFFFFFFFFu >> local))) : record.sample_indices;
					const uint64_t key_frame_bit = uint64_t(record.animated_offset) * 8 + uint64_t(stored_ordinal) * record.pose_bit_size;
					for (uint32_t a = 0; a < num_animated && short_exact_math; ++a)
					{
						const clip_range_entry& range = clip_ranges[a];
						if (range.quad_index != range.track_index * 3)
							continue;
						const plan_entry& entry = plan[size_t(si) * num_animated + a];
						const uint32_t num_bits = entry.bit_offset_and_width >> 24;
						if (is_raw_width(num_bits))
							continue;
						float value[3];
						for (uint32_t c = 0; c < 3; ++c)
							value[c] = decoded(entry, range, c, num_bits == 0 ? 0u : read_field(key_frame_bit + (entry.bit_offset_and_width & 0x00FFFFFFu) + uint64_t(c) * num_bits, num_bits));
						// quat_from_positive_w (math/quatf.h:135-147), one rounding per operation
						const float w_squared = std::fabs(((1.0f - value[0] * value[0]) - value[1] * value[1]) - value[2] * value[2]);
						if (w_squared > 0.0f && w_squared < k_gap)
							short_exact_math = false;
					}
				}
			}
			for (uint32_t track = 0; track < num_tracks && short_exact_math; ++track)
			{
				const float* value = &base_pose[size_t(track * 3) * 4];
				const uint32_t marker = reinterpret_cast<const uint32_t*>(value)[3];
				if (!is_special_quad(marker))		// a constant rotation (W rebuilt above with the host's sqrtf)
					short_exact_math = std::fabs(value[0]) <= k_huge && std::fabs(value[1]) <= k_huge && std::fabs(value[2]) <= k_huge;
			}
		}
		return { short_exact_math, raw_rotations };
	}
}

namespace
{
	uint32_t database_clip_segments(const host_database& db, uint32_t clip_header_offset);		// host_databases.inl

	// The optional metadata behind a transform clip's compressed data (optional_metadata_header, acl_format.h): which sections the blob
	// stores, its parent indices (compressed_tracks::get_parent_track_index, core/impl/compressed_tracks.impl.h:175-190) and track
	// descriptions (get_track_description, :214-275) as [num_tracks][14] = default_value (12 floats, a row of default_values) | precision |
	// shell_distance. A section whose offset does not keep it inside the blob counts as not stored: the decode never reads any of this.
	void parse_clip_metadata(const uint8_t* blob, uint32_t blob_size, const tracks_header& header, aclhip_clip_metadata_info& out_info, std::vector<uint32_t>& out_parents, std::vector<float>& out_descriptions)
	{
		out_info = {};
		out_parents.clear();
		out_descriptions.clear();
		if (!header.has_metadata() || blob_size < k_transform_header_offset + sizeof(optional_metadata_header))
			return;
		out_info.has_metadata = 1;
		optional_metadata_header metadata;
		std::memcpy(&metadata, blob + blob_size - sizeof(metadata), sizeof(metadata));
		const uint64_t limit = blob_size - sizeof(metadata);
		const uint32_t num_tracks = header.num_tracks;
		const auto inside = [&](uint32_t offset, uint64_t bytes, uint32_t alignment) { return offset != k_invalid_offset && offset % alignment == 0 && uint64_t(offset) + bytes <= limit; };

		out_info.has_track_list_name = inside(metadata.track_list_name, 1, 1) ? 1 : 0;
		out_info.has_track_names = inside(metadata.track_name_offsets, uint64_t(num_tracks) * 4, 4) ? 1 : 0;
		out_info.has_contributing_error = inside(metadata.contributing_error, 1, 1) ? 1 : 0;
		if (inside(metadata.parent_track_indices, uint64_t(num_tracks) * 4, 4))
		{
			out_info.has_parent_track_indices = 1;
			out_parents.resize(num_tracks);
			if (num_tracks != 0)
				std::memcpy(out_parents.data(), blob + metadata.parent_track_indices, size_t(num_tracks) * 4);
		}
		// transform descriptions: precision, shell_distance, [3 constant thresholds, v02_00 .. v02_01_99], [default_value: 10 floats, from v02_01_99 on]
		const uint32_t floats = 5u + (header.version >= k_version_v02_01_99 ? 10u : 0u) - (header.version >= k_version_v02_01_99_1 ? 3u : 0u);
		if (header.track_type == k_track_type_qvvf && out_info.has_parent_track_indices != 0 && inside(metadata.track_descriptions, uint64_t(num_tracks) * floats * 4, 4))
		{
			out_info.has_track_descriptions = 1;
			out_descriptions.assign(size_t(num_tracks) * 14, 0.0f);
			for (uint32_t track = 0; track < num_tracks; ++track)
			{
				float data[15];
				std::memcpy(data, blob + metadata.track_descriptions + size_t(track) * floats * 4, size_t(floats) * 4);
				float* row = &out_descriptions[size_t(track) * 14];
				row[3] = 1.0f; row[8] = 1.0f; row[9] = 1.0f; row[10] = 1.0f;		// qvv_identity (before v02_01_99 there is no default_value)
				if (header.version >= k_version_v02_01_99)
				{
					const float* value = data + 2 + (header.version < k_version_v02_01_99_1 ? 3 : 0);
					std::memcpy(row + 0, value + 0, 16);		// rotation xyzw
					std::memcpy(row + 4, value + 4, 12);		// translation xyz
					std::memcpy(row + 8, value + 7, 12);		// scale xyz
				}
				row[12] = data[0];
				row[13] = data[1];
			}
		}
	}
}

static aclhip_status register_clip_impl(aclhip_context* context, const void* compressed_tracks, uint64_t size, int check_hash, aclhip_database database, aclhip_clip* out_clip,
	bool validate_only = false, uint32_t* out_facts = nullptr)
{
	if (out_facts != nullptr)
		*out_facts = 0;
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
	// the packed formats: only the variable ones carry a format byte per segment, segment ranges and a clip range (validate_clip)
	const bool rotations_variable = num_tracks == 0 || header.rotation_format() == k_rotation_quatf_drop_w_variable;
	const bool rotations_full = num_tracks != 0 && header.rotation_format() == k_rotation_quatf_full;
	const bool translations_variable = num_tracks == 0 || header.translation_format() == k_vector_vector3f_variable;
	const bool scales_variable = num_tracks == 0 || header.scale_format() == k_vector_vector3f_variable;
	const uint32_t num_rotations_padded = rotations_variable ? align_to_u32(num_animated_rotations, 4) : 0u;
	const uint32_t num_variable_translations = translations_variable ? num_animated_translations : 0u;
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
	if (num_segments > (1u << 27))
		return fail(context, ACLHIP_ERROR_UNSUPPORTED_FORMAT, "too many segments");
	bool has_raw = false;
	std::vector<uint32_t> segment_pose_bit_sizes(num_segments, 0u);

	if (num_tracks != 0)
	{
		const uint32_t num_entries = (num_tracks + 15) / 16;
		const uint32_t* types = reinterpret_cast<const uint32_t*>(tbase + th.sub_track_types_offset);
		const float* constant_rotations = reinterpret_cast<const float*>(tbase + th.constant_track_data_offset);
		const float* constant_translations = constant_rotations + size_t(th.num_constant_rotation_samples) * (rotations_full ? 4 : 3);
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

					if (kind == 0 && rotations_full)
					{
						// unpack_quat_128 (constant_track_cache.transform.h:136-149): AOS xyzw, stored whole -- W as written, no reconstruction
						std::memcpy(value, constant_rotations + size_t(index) * 4, 16);
					}
					else if (kind == 0)
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
					if (is_special_quad(value_bits[3]))
						value_bits[3] &= 0x7FFFFFFFu;	// only a garbage (NaN) constant could collide with the markers
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
		// (equal, not "at most": the reference sizes its SOA groups of constant rotations and finds the translations and scales behind them
		// from the header's counts, decompression.transform.h -- types that name fewer constants than the header decode differently there)
		if (constant_counts[0] != constant_limits[0] || constant_counts[1] != constant_limits[1] || (has_scale && constant_counts[2] != constant_limits[2]))
			return fail(context, ACLHIP_ERROR_INVALID_CLIP, "sub-track types disagree with the constant sample counts");

		// clip ranges: rotations are SOA per group of 4 (last group unpadded), translations / scales AOS (write_range_data.h:79-207)
		{
			const float* range_data = reinterpret_cast<const float*>(tbase + th.clip_range_data_offset);
			// (sub-tracks of a full format have no clip range: min 0 / extent 1, never used -- their samples are raw)
			for (uint32_t i = 0; i < num_animated; ++i)
				for (uint32_t c = 0; c < 3; ++c)
				{
					clip_ranges[i].range_min[c] = 0.0f;
					clip_ranges[i].range_extent[c] = 1.0f;
				}
			for (uint32_t i = 0; rotations_variable && i < num_animated_rotations; ++i)
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
			const float* vector_ranges = range_data + size_t(rotations_variable ? num_animated_rotations : 0u) * 6;
			for (uint32_t i = num_animated_rotations; i < num_animated; ++i)
			{
				const bool is_translation = i < num_animated_rotations + num_animated_translations;
				if (!(is_translation ? translations_variable : scales_variable))
					continue;
				const uint32_t range_index = is_translation ? i - num_animated_rotations : num_variable_translations + (i - num_animated_rotations - num_animated_translations);
				const float* entry = vector_ranges + size_t(range_index) * 6;
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
			if (start >= end || end > num_samples || (si == 0 && start != 0))		// (validate_clip bounds a segment's length)
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
			segment_pose_bit_sizes[si] = sh.animated_pose_bit_size;
			for (uint32_t sample = start; sample < end; ++sample)
			{
				// (sample_record: the keyframes the segment keeps, or -- nothing stripped -- the sample's index inside its segment, of any length)
				record.sample_indices = stripped ? reinterpret_cast<const stripped_segment_header&>(sh).sample_indices : sample - start;
				record.segment_and_local = (si << 5) | ((sample - start) & 31u);
				samples[sample] = record;
			}

			// (validate_clip checked these offsets in 64 bit arithmetic, and that every keyframe the clip stores lies inside the blob)

			uint32_t bit_offset = 0;
			for (uint32_t a = 0; a < num_animated; ++a)
			{
				const bool is_rotation = a < num_animated_rotations;
				const bool is_translation = !is_rotation && a < num_animated_rotations + num_animated_translations;
				const bool is_variable = is_rotation ? rotations_variable : (is_translation ? translations_variable : scales_variable);
				// ordinal among the sub-tracks that HAVE metadata, translations and scales behind the (padded) rotations
				const uint32_t vector_index = is_translation ? a - num_animated_rotations : num_variable_translations + (a - num_animated_rotations - num_animated_translations);
				const uint32_t format_index = is_rotation ? a : num_rotations_padded + vector_index;
				// full formats: every sample raw, 96 bits -- 128 for quatf_full -- and no format byte (animated_track_cache.transform.h:608-620,921-926)
				const uint32_t stored_bits = is_variable ? format_per_track[format_index] : raw_num_bits;
				const bool is_raw = stored_bits == raw_num_bits;
				if (!is_raw && stored_bits > 23)
					return fail(context, ACLHIP_ERROR_INVALID_CLIP, "segment %u sub-track %u has an invalid bit width %u", si, a, stored_bits);
				if (bit_offset > k_quad_ordinal_mask)
					return fail(context, ACLHIP_ERROR_UNSUPPORTED_FORMAT, "keyframes larger than 2 MiB are not supported");

				plan_entry& entry = plan[size_t(si) * num_animated + a];
				const uint32_t num_bits = !is_raw ? stored_bits : (is_variable ? k_width_raw_variable : (is_rotation && rotations_full ? k_width_raw_quat : k_width_raw_full));
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

				bit_offset += stored_sample_bits(num_bits);
				// The reference finds a keyframe's translations behind animated_rotation_bit_size bits and its scales behind
				// animated_translation_bit_size more (decompression.transform.h:533-536, animated_track_cache.transform.h): the kernels here
				// take every sub-track's position from the widths in front of it and never read the two fields -- a blob in which they
				// disagree with the widths would decode HERE and send the reference's walk anywhere. Refused, like the pose size below.
				// (found with the oracle under AddressSanitizer on blobs the validators had accepted, round 5)
				if (a + 1 == num_animated_rotations && bit_offset != sh.animated_rotation_bit_size)
					return fail(context, ACLHIP_ERROR_INVALID_CLIP, "segment %u: rotation widths add up to %u bits, header says %u", si, bit_offset, sh.animated_rotation_bit_size);
				if (a + 1 == num_animated_rotations + num_animated_translations && bit_offset != sh.animated_rotation_bit_size + sh.animated_translation_bit_size)
					return fail(context, ACLHIP_ERROR_INVALID_CLIP, "segment %u: translation widths add up to %u bits, header says %u", si, bit_offset - sh.animated_rotation_bit_size, sh.animated_translation_bit_size);
			}
			if (num_animated_rotations == 0 && sh.animated_rotation_bit_size != 0)
				return fail(context, ACLHIP_ERROR_INVALID_CLIP, "segment %u: no animated rotations, header says %u bits of them", si, sh.animated_rotation_bit_size);
			if (num_animated_translations == 0 && sh.animated_translation_bit_size != 0)
				return fail(context, ACLHIP_ERROR_INVALID_CLIP, "segment %u: no animated translations, header says %u bits of them", si, sh.animated_translation_bit_size);

			if (bit_offset != sh.animated_pose_bit_size)
				return fail(context, ACLHIP_ERROR_INVALID_CLIP, "segment %u: sub-track widths add up to %u bits, header says %u", si, bit_offset, sh.animated_pose_bit_size);
		}
	}

	// ---- would the reference find the same segments? ----
	// The reference has no table: it guesses a key's segment from num_samples / num_segments and scans up to four start indices from the
	// segment in front of the guess (decompression.transform.h:372-409). For every cut its compressor makes that lands on the segment the
	// sample records name; for start indices edited by hand (or a sample count that disagrees with them) it may not -- such a blob would
	// decode to other poses here than there. Refused. (Found by decoding mutated-but-accepted clips on the GPU against the oracle, round 5.)
	if (multi_segment && num_tracks != 0)
	{
		const uint32_t* starts = reinterpret_cast<const uint32_t*>(tbase + k_segment_start_indices_offset);		// num_segments + 1 entries, the last one 0xFFFFFFFF (validate_clip)
		const uint32_t approx_samples_per_segment = num_samples / num_segments;
		for (uint32_t sample = 0; sample < num_samples; ++sample)
		{
			const uint32_t approx_segment = approx_samples_per_segment != 0 ? sample / approx_samples_per_segment : 0xFFFFFFFFu;
			const uint32_t first = approx_segment > 0 ? approx_segment - 1 : 0;
			uint32_t found = 0xFFFFFFFFu;
			for (uint32_t segment = first; segment < first + 4 && segment <= num_segments; ++segment)
				if (sample < starts[segment])
				{
					found = segment - 1;		// (segment == 0 cannot match: starts[0] is 0)
					break;
				}
			if (approx_samples_per_segment == 0 || found != (samples[sample].segment_and_local >> 5))
				return fail(context, ACLHIP_ERROR_INVALID_CLIP, "sample %u: the reference's segment lookup would not find segment %u", sample, samples[sample].segment_and_local >> 5);
		}
	}

	// ---- animated sub-tracks in POSE order ----
	// The tables above follow the bitstream (rotations, translations, scales); lanes do not care which sub-track they get, so the
	// tables are reordered by destination window (and by kind inside a window). The sub-tracks that land in quads [c * k_image_chunk_quads, (c + 1) * ..) are then
	// a contiguous range of ordinals, image_chunks[c] .. image_chunks[c + 1]: the pose kernel can build a pose of any size through
	// a fixed LDS window.
	const uint32_t num_image_chunks = num_pose_windows(num_tracks);
	std::vector<uint32_t> image_chunks(window_spans_word_offset(num_image_chunks), num_animated);
	if (num_animated != 0)
	{
		std::vector<uint32_t> order(num_animated);		// new ordinal -> bitstream ordinal
		for (uint32_t a = 0; a < num_animated; ++a)
			order[a] = a;
		// inside a window rotations come first: the rotation math (two square roots, a division) is most of a lane's work and every
		// pass that holds a rotation pays for it, so rotations are packed into as few passes as possible
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

#if defined(ACLHIP_EXPERIMENTS)
	// ---- where a window's bits sit inside a keyframe (the staged kernel of kernels_experiments.inl) ----
	// Per (segment, window) and kind: the run of keyframe bits its animated sub-tracks cover (window_span_entry); per clip, what a
	// wave needs in LDS to stage one keyframe's runs (16 byte pieces, any alignment of the keyframe) and to keep a window's decoded
	// sub-tracks.
	std::vector<window_span_entry> window_spans(std::max<size_t>(size_t(num_segments) * num_image_chunks, 1));
	std::memset(window_spans.data(), 0, window_spans.size() * sizeof(window_span_entry));
	uint32_t window_animated_max = 0, window_key_bytes_max = 0;
	for (uint32_t chunk = 0; chunk < num_image_chunks && num_animated != 0; ++chunk)
	{
		const uint32_t first = image_chunks[chunk], end = chunk + 1 < num_image_chunks ? image_chunks[chunk + 1] : num_animated;
		window_animated_max = std::max(window_animated_max, end - first);
		for (uint32_t si = 0; si < num_segments; ++si)
		{
			window_span_entry& span = window_spans[size_t(si) * num_image_chunks + chunk];
			bool seen[3] = { false, false, false };
			for (uint32_t a = first; a < end; ++a)
			{
				const plan_entry& entry = plan[size_t(si) * num_animated + a];
				const uint32_t width = entry.bit_offset_and_width >> 24;
				if (width == 0)
					continue;		// constant in this segment: nothing is read
				const uint32_t kind = clip_ranges[a].quad_index - clip_ranges[a].track_index * 3;
				const uint32_t bit = entry.bit_offset_and_width & 0x00FFFFFFu;
				span.first_bit[kind] = seen[kind] ? std::min(span.first_bit[kind], bit) : bit;
				span.end_bit[kind] = seen[kind] ? std::max(span.end_bit[kind], bit + stored_sample_bits(width)) : bit + stored_sample_bits(width);
				seen[kind] = true;
			}
			uint32_t key_bytes = 0;
			for (uint32_t kind = 0; kind < 3; ++kind)
				if (seen[kind])
					key_bytes += staged_run_bytes(span.end_bit[kind] - span.first_bit[kind]);
			window_key_bytes_max = std::max(window_key_bytes_max, key_bytes);
		}
	}

#endif

	// ---- can a scale of this clip come out negative? ----
	// rtm::qvv_mul composes matrices instead of quaternions when a scale component of either operand is negative; the pose consumers
	// compile that route in only while a registered clip can get there (kernels_consumers.inl: kMirrored). Conservative: a constant or
	// default scale below zero (or not a number), an animated scale whose clip range reaches below zero -- a decoded value is
	// clip_min + clip_extent * [0, 1] --, or one that is stored raw in some segment (raw samples bypass the ranges).
	bool negative_scale_possible = false;
	if (num_tracks != 0)
	{
		const auto may_be_negative = [](float value) { return !(value >= 0.0f); };
		if (!has_scale)
			negative_scale_possible = may_be_negative(float(header.default_scale()));
		for (uint32_t track = 0; has_scale && track < num_tracks; ++track)
		{
			const float* value = &base_pose[size_t(track * 3 + 2) * 4];
			const uint32_t marker = reinterpret_cast<const uint32_t*>(value)[3];
			if (!is_special_quad(marker) || (marker & k_quad_animated) == 0)		// constant, or default (the base pose holds the default's xyz)
				negative_scale_possible = negative_scale_possible || may_be_negative(value[0]) || may_be_negative(value[1]) || may_be_negative(value[2]);
		}
		for (uint32_t a = 0; a < num_animated; ++a)
		{
			const clip_range_entry& range = clip_ranges[a];
			if (range.quad_index != range.track_index * 3 + 2)
				continue;
			for (uint32_t c = 0; c < 3; ++c)
				negative_scale_possible = negative_scale_possible || may_be_negative(range.range_min[c]) || may_be_negative(range.range_min[c] + std::min(range.range_extent[c], 0.0f) * 1.01f);
			for (uint32_t si = 0; si < num_segments; ++si)
				negative_scale_possible = negative_scale_possible || is_raw_width(plan[size_t(si) * num_animated + a].bit_offset_and_width >> 24);
		}
	}

	// ---- may the kernels use the SHORT correctly rounded square root / reciprocal on this clip's rotations? (analyze_rotation_values above) ----
	const rotation_value_facts rotation_facts = analyze_rotation_values(blob, blob_size, header.has_database(), stripped, num_tracks, num_samples, num_segments, num_animated, plan, clip_ranges, samples, base_pose);
	// (quatf_full: a stored W is any float, constants included -- nothing the analysis proves about x, y, z bounds the norms)
	bool short_exact_math = rotation_facts.short_exact_math && !rotations_full;
	const bool raw_rotations = rotation_facts.raw_rotations || rotations_full;
	// ACLHIP_SHORT_EXACT_MATH = 0: never (A/B measurements; harmless: the compiler's forms are exact everywhere). = 1: ALWAYS, whatever
	// the analysis says -- it BREAKS bit exactness on the clips the analysis exists for, so only a lab build (-DACLHIP_LAB_KNOBS:
	// libaclhip_lab.so, what tests/test_gpu_exact_math.py uses to show the analysis has teeth) listens to it.
	static const int short_exact_off = []() { const char* value = path_knob("ACLHIP_SHORT_EXACT_MATH"); return value != nullptr && value[0] == '0' ? 1 : 0; }();
	static const int short_exact_forced = []() { const char* value = lab_knob("ACLHIP_SHORT_EXACT_MATH"); return value != nullptr && value[0] == '1' ? 1 : 0; }();
	if (short_exact_off != 0)
		short_exact_math = false;
	if (short_exact_forced != 0)
		short_exact_math = true;

	// ---- one device allocation: blob (+ zeroed tail padding) | base pose | segments | plan | clip ranges | sample -> segment ----
	const uint64_t blob_bytes = align_to_u32(blob_size, 16) + 64;		// windows of up to 16 bytes are read: keep well past the reference's 15 bytes of slack
	const uint64_t base_pose_offset = blob_bytes;
	// resolved pose: what a decode with the track_writer defaults stores for every non animated sub-track (animated slots: zero)
	std::vector<float> resolved_pose(base_pose);
	for (uint32_t quad = 0; quad < num_quads; ++quad)
	{
		uint32_t* value_bits = reinterpret_cast<uint32_t*>(&resolved_pose[size_t(quad) * 4]);
		if (is_special_quad(value_bits[3]))
			resolved_pose[size_t(quad) * 4 + 3] = (value_bits[3] & (k_quad_animated | k_quad_default_w_one)) == k_quad_default_w_one ? 1.0f : 0.0f;
	}

	const uint64_t resolved_pose_offset = base_pose_offset + uint64_t(num_quads) * 16;
	// the resolved pose once more as 10 packed floats per track (rotation xyzw | translation xyz | scale xyz), directly behind the first:
	// what the QVV40 output layout starts from (kernels_pose.inl)
	const uint64_t resolved_qvv40_offset = resolved_pose_offset + uint64_t(num_quads) * 16;
	std::vector<float> resolved_qvv40((size_t(num_tracks) * 10 + 3) / 4 * 4 + 4, 0.0f);
	for (uint32_t track = 0; track < num_tracks; ++track)
	{
		const float* record = &resolved_pose[size_t(track) * 12];
		float* packed = &resolved_qvv40[size_t(track) * 10];
		std::memcpy(packed, record, 16);
		std::memcpy(packed + 4, record + 4, 12);
		std::memcpy(packed + 7, record + 8, 12);
	}
	// ... and behind that the clip's bind pose, 12 floats per track (bind_pose_of, aclhip_device.h): ACLHIP_DEFAULT_BIND_POSE
	aclhip_clip_metadata_info metadata_info;
	std::vector<uint32_t> metadata_parents;
	std::vector<float> metadata_descriptions;
	parse_clip_metadata(blob, blob_size, header, metadata_info, metadata_parents, metadata_descriptions);
	static_assert(sizeof(float) == 4, "layout");
	if (resolved_qvv40.size() != resolved_qvv40_floats(num_tracks))
		return fail(context, ACLHIP_ERROR_DEVICE, "internal: packed pose size");
	const uint64_t bind_pose_offset = resolved_qvv40_offset + resolved_qvv40.size() * sizeof(float);
	std::vector<float> bind_pose(std::max<size_t>(size_t(num_tracks) * 12, 4), 0.0f);
	for (uint32_t track = 0; track < num_tracks; ++track)
	{
		float* row = &bind_pose[size_t(track) * 12];
		if (metadata_info.has_track_descriptions != 0)
			std::memcpy(row, &metadata_descriptions[size_t(track) * 14], 48);
		else
		{
			row[3] = 1.0f; row[8] = 1.0f; row[9] = 1.0f; row[10] = 1.0f;		// track_desc_transformf::default_value = qvv_identity (core/track_desc.h:103)
		}
		row[7] = 0.0f;
		row[11] = 0.0f;
	}
	const uint64_t samples_offset = (bind_pose_offset + bind_pose.size() * sizeof(float) + 31) & ~uint64_t(31);
	// clips bound to a database carry a copy of their segments' tier metadata per sample (database_sample_record; zero = not resident
	// until refresh_database_sample_tiers_kernel has run for the clip, below)
	const bool database_samples = database != ACLHIP_INVALID_HANDLE && num_tracks != 0 && header.has_database();
	const size_t sample_record_size = database_samples ? sizeof(database_sample_record) : sizeof(sample_record);
	const uint64_t plan_offset = (samples_offset + samples.size() * sample_record_size + 31) & ~uint64_t(31);		// 32 byte entries from here on
	// (the clip range table DIRECTLY behind the plan: decompress_track_kernel addresses a request's plan rows backwards from it, kernels_track.inl)
	const uint64_t clip_ranges_offset = plan_offset + plan.size() * sizeof(plan_entry);
	const uint64_t image_chunks_offset = clip_ranges_offset + clip_ranges.size() * sizeof(clip_range_entry);
#if defined(ACLHIP_EXPERIMENTS)
	const uint64_t window_spans_offset = image_chunks_offset + image_chunks.size() * sizeof(uint32_t);		// (image_chunks_offset is a multiple of 32: whole 32 byte entries before it)
	const uint64_t total_bytes = window_spans_offset + window_spans.size() * sizeof(window_span_entry);
#else
	const uint64_t total_bytes = image_chunks_offset + image_chunks.size() * sizeof(uint32_t);
#endif
	if (total_bytes > 0xFFFFFFFFull)
		return fail(context, ACLHIP_ERROR_UNSUPPORTED_FORMAT, "clip tables beyond 4 GiB are not supported");

	std::vector<uint8_t> staging(total_bytes, 0);
	std::memcpy(staging.data(), blob, blob_size);
	if (num_quads != 0)
		std::memcpy(staging.data() + base_pose_offset, base_pose.data(), size_t(num_quads) * 16);
	if (num_quads != 0)
		std::memcpy(staging.data() + resolved_pose_offset, resolved_pose.data(), size_t(num_quads) * 16);
	std::memcpy(staging.data() + resolved_qvv40_offset, resolved_qvv40.data(), resolved_qvv40.size() * sizeof(float));
	std::memcpy(staging.data() + bind_pose_offset, bind_pose.data(), bind_pose.size() * sizeof(float));
	if (database_samples)
		for (size_t sample = 0; sample < samples.size(); ++sample)
			std::memcpy(staging.data() + samples_offset + sample * sizeof(database_sample_record), &samples[sample], sizeof(sample_record));		// (tier metadata: the staging bytes are zero)
	else
		std::memcpy(staging.data() + samples_offset, samples.data(), samples.size() * sizeof(sample_record));
	std::memcpy(staging.data() + plan_offset, plan.data(), plan.size() * sizeof(plan_entry));
	std::memcpy(staging.data() + clip_ranges_offset, clip_ranges.data(), clip_ranges.size() * sizeof(clip_range_entry));
	std::memcpy(staging.data() + image_chunks_offset, image_chunks.data(), image_chunks.size() * sizeof(uint32_t));
#if defined(ACLHIP_EXPERIMENTS)
	std::memcpy(staging.data() + window_spans_offset, window_spans.data(), window_spans.size() * sizeof(window_span_entry));
#endif
	// aclhip_analyze_clip: what registration derives about the clip's VALUES (the kernels' variants follow from these)
	if (out_facts != nullptr)
		*out_facts = (rotation_facts.short_exact_math ? ACLHIP_CLIP_FACT_SHORT_EXACT_MATH : 0u) | (raw_rotations ? ACLHIP_CLIP_FACT_RAW_ROTATIONS : 0u)		// (the analysis, not what a knob made of it)
			| (negative_scale_possible ? ACLHIP_CLIP_FACT_NEGATIVE_SCALE : 0u);
	if (validate_only)
		return ACLHIP_OK;		// aclhip_check_clip: everything above is host work

	std::lock_guard<std::shared_mutex> lock(context->mutex);
	device_guard guard(context->device);
	if (!guard.ok)
		return fail(context, ACLHIP_ERROR_DEVICE, "hipSetDevice(%d) failed", context->device);
	collect_retired(context, false);

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
		record.flags |= (has_scale || float(header.default_scale()) != 1.0f) ? k_clip_scaled : 0u;
		record.flags |= stripped ? k_clip_has_stripped_keyframes : 0u;
		record.flags |= header.has_database() ? k_clip_has_database : 0u;
		record.flags |= (header.version > k_version_first && header.is_wrap_optimized()) ? k_clip_wraps : 0u;
		record.flags |= has_raw ? k_clip_has_raw : 0u;
		record.flags |= negative_scale_possible ? k_clip_negative_scale : 0u;
		record.flags |= short_exact_math ? k_clip_short_exact_math : 0u;
		record.flags |= raw_rotations ? k_clip_raw_rotations : 0u;
		record.flags |= rotations_full ? k_clip_full_rotations : 0u;
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
			// ... and the database keeps exactly this clip's segments behind its runtime clip header (a metadata offset moved by one
			// segment header leaves the clip in front one header short: its last segment's tier words would be the next clip's hash)
			contained = contained && database_clip_segments(db, record.db_clip_header_offset) == record.num_segments;
		}
		if (!contained)
		{
			free_clip_memory(context, d_memory);
			context->free_slots.push_back(slot);
			return fail(context, ACLHIP_ERROR_NOT_IN_DATABASE, "the database does not contain this clip");
		}
		// every keyframe a tier holds for this clip must lie inside that tier's bulk data: the chunk segment headers only carry an
		// offset, the size of a keyframe is the clip's (animated_pose_bit_size of the segment)
		for (uint32_t si = 0; contained && si < num_segments; ++si)
		{
			const uint32_t segment_header_offset = record.db_clip_header_offset + uint32_t(sizeof(database_runtime_clip_header)) + si * uint32_t(sizeof(database_runtime_segment_header));
			const uint64_t pose_bit_size = segment_pose_bit_sizes[si];
			for (int tier = 0; tier < 2; ++tier)
			{
				const std::vector<tier_patch>& patches = db.patches_by_header[tier];
				auto patch = std::lower_bound(patches.begin(), patches.end(), segment_header_offset,
					[](const tier_patch& entry, uint32_t offset) { return entry.segment_header_offset < offset; });
				for (; patch != patches.end() && patch->segment_header_offset == segment_header_offset; ++patch)
					if (uint64_t(patch->samples_offset) + (uint64_t(__builtin_popcount(patch->sample_indices)) * pose_bit_size + 7) / 8 > db.info.bulk_data_size[tier])
						contained = false;
			}
		}
		if (!contained)
		{
			free_clip_memory(context, d_memory);
			context->free_slots.push_back(slot);
			return fail(context, ACLHIP_ERROR_INVALID_CLIP, "the database's keyframes for this clip lie outside of its bulk data");
		}
		if (db.streamed)
			for (uint32_t si = 0; si < num_segments; ++si)
				db.segment_pose_bits.emplace_back(record.db_clip_header_offset + uint32_t(sizeof(database_runtime_clip_header)) + si * uint32_t(sizeof(database_runtime_segment_header)), segment_pose_bit_sizes[si]);
		record.db_headers = db.d_runtime_headers;
		record.db_bulk_data[0] = db.d_bulk_data[0];
		record.db_bulk_data[1] = db.d_bulk_data[1];
		record.flags |= k_clip_database_samples;
	}

	size_t staging_used = 0;
	if (!stage_upload(context, d_memory, staging.data(), total_bytes, staging_used)
		|| !stage_upload(context, context->d_clips + slot, &record, sizeof(record), staging_used) || !finish_uploads(context))
	{
		free_clip_memory(context, d_memory);
		context->free_slots.push_back(slot);
		return fail(context, ACLHIP_ERROR_DEVICE, "uploading the clip failed");
	}
	context->clips_registered++;
	if (database != ACLHIP_INVALID_HANDLE)
	{
		// the database's list of bound clips grows by this one, and the clip's sample records get the tiers' current state: on the copy
		// stream, BEHIND everything already enqueued on the streams this context launched on (a stream_in that has not executed yet
		// must not be overtaken: its own refresh ran over a list without this clip). Binding a clip to a database is the one
		// registration that waits for work in flight.
		host_database& db = context->databases[database];
		db.num_bound_clips++;
		db.bound_clips.push_back(slot);
		const aclhip_status list_status = upload_bound_clips(context, db);
		bool refreshed = list_status == ACLHIP_OK && order_stream_behind_launches(context, context->copy_stream);
		if (refreshed)
		{
			hipLaunchKernelGGL(refresh_database_sample_tiers_kernel, dim3(1), dim3(256), 0, context->copy_stream,
				context->d_clips, context->d_clips_capacity, db.d_bound_clips + (db.bound_clips.size() - 1), 1u, db.d_runtime_headers);
			refreshed = hipGetLastError() == hipSuccess && finish_uploads(context);
		}
		if (!refreshed)
		{
			db.bound_clips.pop_back();
			db.num_bound_clips--;
			device_clip cleared;
			std::memset(&cleared, 0, sizeof(cleared));
			size_t cleared_staging = 0;
			(void)stage_upload(context, context->d_clips + slot, &cleared, sizeof(cleared), cleared_staging);
			(void)finish_uploads(context);
			free_clip_memory(context, d_memory);
			context->free_slots.push_back(slot);
			return list_status != ACLHIP_OK ? list_status : fail(context, ACLHIP_ERROR_DEVICE, "binding the clip to the database failed");
		}
	}

	host_clip& entry = context->clips[slot];
	entry.in_use = true;
	entry.database = database;
	if (database != ACLHIP_INVALID_HANDLE && context->databases[database].streamed)
	{
		entry.db_first_segment_header = record.db_clip_header_offset + uint32_t(sizeof(database_runtime_clip_header));
		entry.db_num_segments = num_segments;
	}
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
	entry.pose_quads = num_quads;
	context->max_pose_quads = std::max(context->max_pose_quads, num_quads);
#if defined(ACLHIP_EXPERIMENTS)
	entry.window_animated = window_animated_max;
	entry.window_key_bytes = window_key_bytes_max;
	context->max_window_animated = std::max(context->max_window_animated, window_animated_max);
	context->max_window_key_bytes = std::max(context->max_window_key_bytes, window_key_bytes_max);
#endif
	entry.metadata = metadata_info;
	entry.metadata_parents = std::move(metadata_parents);
	entry.metadata_descriptions = std::move(metadata_descriptions);
	entry.scaled = num_tracks != 0 && (has_scale || float(header.default_scale()) != 1.0f);
	context->num_scaled_clips += entry.scaled ? 1u : 0u;
	entry.negative_scale = negative_scale_possible;
	context->num_negative_scale_clips += entry.negative_scale ? 1u : 0u;

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

	std::lock_guard<std::shared_mutex> lock(context->mutex);
	if (clip >= context->clips.size() || !context->clips[clip].in_use)
		return fail(context, ACLHIP_ERROR_UNKNOWN_CLIP, "unknown clip handle %u", clip);

	device_guard guard(context->device);
	collect_retired(context, false);

	// Everything is stream ordered and nobody waits here. Decodes ALREADY ENQUEUED on the streams this context launched on still
	// find the clip -- a kernel reads the table record when it executes, so the record is cleared (on the context's own retire stream)
	// only behind those launches --; launches that execute after that are refused (counted, their poses untouched); the clip's memory,
	// its share of a hierarchy image and the handle itself are given back once both have happened. Callers still owe the reference's
	// contract: no decode of a clip is ENQUEUED after its unregistration.
	{
		aclhip_context::retired_item item;
		item.clip_memory = context->clips[clip].device_memory;
		item.hierarchy = context->clips[clip].d_hierarchy;
		item.slot = clip;
		retire(context, std::move(item), context->d_clips + clip);
	}
	context->clips_unregistered++;
	const uint32_t bound_database = context->clips[clip].database;
	if (bound_database != ACLHIP_INVALID_HANDLE && bound_database < context->databases.size() && context->databases[bound_database].num_bound_clips != 0)
	{
		host_database& db = context->databases[bound_database];
		db.num_bound_clips--;
		// (the device copy of the list keeps the handle until the next bind: refreshes skip entries whose table record is no longer a
		// clip of this database)
		for (size_t i = 0; i < db.bound_clips.size(); ++i)
			if (db.bound_clips[i] == clip)
			{
				db.bound_clips[i] = db.bound_clips.back();
				db.bound_clips.pop_back();
				break;
			}
		(void)upload_bound_clips(context, db);
		// the clip's runtime segment headers no longer constrain chunks that arrive (streamed databases)
		const uint32_t first = context->clips[clip].db_first_segment_header, count = context->clips[clip].db_num_segments;
		for (uint32_t si = 0; si < count; ++si)
		{
			const uint32_t offset = first + si * uint32_t(sizeof(database_runtime_segment_header));
			for (size_t i = 0; i < db.segment_pose_bits.size(); ++i)
				if (db.segment_pose_bits[i].first == offset)
				{
					db.segment_pose_bits[i] = db.segment_pose_bits.back();
					db.segment_pose_bits.pop_back();
					break;
				}
		}
	}
	const host_clip removed = context->clips[clip];
	context->clips[clip] = host_clip();
	context->num_scaled_clips -= removed.scaled ? 1u : 0u;
	context->num_negative_scale_clips -= removed.negative_scale ? 1u : 0u;
	context->num_wide_scalar_clips -= removed.wide_scalar ? 1u : 0u;
	if ((removed.pose_quads != 0 && removed.pose_quads == context->max_pose_quads) || (removed.hierarchy_words != 0 && removed.hierarchy_words == context->max_hierarchy_words)
		|| (removed.window_animated != 0 && removed.window_animated == context->max_window_animated) || (removed.window_key_bytes != 0 && removed.window_key_bytes == context->max_window_key_bytes)
		|| (removed.scalar_tracks != 0 && removed.scalar_tracks == context->max_scalar_tracks) || (removed.scalar_frame_bytes != 0 && removed.scalar_frame_bytes == context->max_scalar_frame_bytes))
		recompute_launch_maxima(context);
	return ACLHIP_OK;
}
