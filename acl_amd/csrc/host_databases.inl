// host_databases.inl -- part of aclhip.hip (one translation unit; included there, in this order, not compiled on its own).
// Host side: compressed_database registration and streaming, host only validation entry points, clip queries.

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
		(void)hipFree(db.d_bound_clips);
		for (int tier = 0; tier < 2; ++tier)
		{
			(void)hipFree(db.d_bulk_data[tier]);
			(void)hipFree(db.d_patches[tier]);
			if (db.pinned_bulk_data[tier] != nullptr)
				(void)hipHostFree(db.pinned_bulk_data[tier]);
			if (db.pinned_patches[tier] != nullptr)
				(void)hipHostFree(db.pinned_patches[tier]);
		}
		db = host_database();
	}
}


namespace
{
	// segments of the clip whose runtime clip header sits at `clip_header_offset` (the clip metadata tile the block: register_database_impl);
	// 0 when no clip's does
	uint32_t database_clip_segments(const host_database& db, uint32_t clip_header_offset)
	{
		const auto found = std::lower_bound(db.clip_metadata.begin(), db.clip_metadata.end(), clip_header_offset,
			[](const database_clip_metadata& metadata, uint32_t offset) { return metadata.clip_header_offset < offset; });
		if (found == db.clip_metadata.end() || found->clip_header_offset != clip_header_offset)
			return 0;
		const uint64_t end = found + 1 != db.clip_metadata.end() ? (found + 1)->clip_header_offset : db.runtime_headers_size;
		return uint32_t((end - clip_header_offset - sizeof(database_runtime_clip_header)) / sizeof(database_runtime_segment_header));
	}

	// a chunk segment header must name one of the segment headers of the clip it names
	bool names_a_segment_header(const host_database& db, uint32_t clip_header_offset, uint32_t segment_header_offset)
	{
		const uint32_t segments = database_clip_segments(db, clip_header_offset);
		const uint64_t first = uint64_t(clip_header_offset) + sizeof(database_runtime_clip_header);
		return segments != 0 && segment_header_offset >= first && (segment_header_offset - first) % sizeof(database_runtime_segment_header) == 0
			&& (segment_header_offset - first) / sizeof(database_runtime_segment_header) < segments;
	}
}

static aclhip_status register_database_impl(aclhip_context* context, const void* compressed_database, uint64_t size,
	const void* bulk_data_medium, const void* bulk_data_low, int check_hash, aclhip_database* out_database, bool validate_only, bool streamed = false)
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
	if ((header.clip_metadata_offset & 3u) != 0)
		return fail(context, ACLHIP_ERROR_INVALID_CLIP, "Header offsets are not 4 byte aligned");

	const bool is_inline = (header.misc_packed & 1u) != 0;
	const uint8_t* bulk_sources[2] = { static_cast<const uint8_t*>(bulk_data_medium), static_cast<const uint8_t*>(bulk_data_low) };
	for (int tier = 0; tier < 2 && !streamed; ++tier)
	{
		if (header.bulk_data_size[tier] == 0)
			continue;
		if (bulk_sources[tier] == nullptr)
		{
			if (!is_inline || header.bulk_data_offset[tier] == k_invalid_offset || uint64_t(header.bulk_data_offset[tier]) + header.bulk_data_size[tier] > header_limit
				|| (header.bulk_data_offset[tier] & 3u) != 0)
				return fail(context, ACLHIP_ERROR_INVALID_ARGUMENT, "tier %d has %u bytes of bulk data: pass it in, it is not inline", tier + 1, header.bulk_data_size[tier]);
			bulk_sources[tier] = hbase + header.bulk_data_offset[tier];
		}
		if (check_hash && hash32(bulk_sources[tier], header.bulk_data_size[tier]) != header.bulk_data_hash[tier])
			return fail(context, ACLHIP_ERROR_INVALID_CLIP, "Invalid bulk data hash (tier %d)", tier + 1);
	}

	host_database db;
	db.streamed = streamed;
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
	// The runtime block is [clip header, one segment header per segment of the clip] per clip, in clip order (build_database,
	// compress.database.impl.h): the clip metadata's offsets start at 0, ascend, leave room for whole segment headers between them and
	// use the block up. A database whose offsets overlap has clip hashes written INTO segment headers by initialize
	// (database.impl.h:151-157) -- the reference, its restatement and this library then each decode the tiers of that segment their own
	// way. Refused. (Found by tools/fuzz_gpu_mutated_db.py, round 6.)
	for (size_t i = 0; i < db.clip_metadata.size(); ++i)
	{
		const database_clip_metadata& metadata = db.clip_metadata[i];
		if (uint64_t(metadata.clip_header_offset) + sizeof(database_runtime_clip_header) > runtime_size || (metadata.clip_header_offset & 7u) != 0)
			return fail(context, ACLHIP_ERROR_INVALID_CLIP, "Clip metadata points outside of the runtime headers");
		const uint64_t end = i + 1 < db.clip_metadata.size() ? db.clip_metadata[i + 1].clip_header_offset : runtime_size;
		const uint64_t first_segment = uint64_t(metadata.clip_header_offset) + sizeof(database_runtime_clip_header);
		if ((i == 0 && metadata.clip_header_offset != 0) || end < first_segment + sizeof(database_runtime_segment_header) || (end - first_segment) % sizeof(database_runtime_segment_header) != 0)
			return fail(context, ACLHIP_ERROR_INVALID_CLIP, "Clip metadata: the runtime headers of clip %zu do not tile the block", i);
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
			if (uint64_t(description.offset) + description.size > header.bulk_data_size[tier] || description.size < sizeof(database_chunk_header) || description.size > header.max_chunk_size
				|| (description.offset & 3u) != 0)
				return fail(context, ACLHIP_ERROR_INVALID_CLIP, "Chunk %u of tier %d lies outside of the bulk data", chunk_index, tier + 1);
			if (streamed)
				continue;		// its header arrives with the chunk (parse_arrived_chunks)
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
				// (8 byte aligned: the device stores the tier words of a segment header as 64 bit atomics)
				if (uint64_t(segments[i].segment_header_offset) + sizeof(database_runtime_segment_header) > runtime_size || segments[i].samples_offset >= header.bulk_data_size[tier]
					|| (segments[i].segment_header_offset & 7u) != 0 || !names_a_segment_header(db, segments[i].clip_header_offset, segments[i].segment_header_offset))
					return fail(context, ACLHIP_ERROR_INVALID_CLIP, "Chunk %u of tier %d points outside of the database", chunk_index, tier + 1);
				patches[tier].push_back(tier_patch{ segments[i].segment_header_offset, segments[i].sample_indices, segments[i].samples_offset });
			}
		}
		if (!streamed)
		{
			db.chunk_first_patch[tier][num_chunks] = uint32_t(patches[tier].size());
			db.num_parsed_chunks[tier] = num_chunks;
		}
		db.patches_by_header[tier] = patches[tier];
		std::sort(db.patches_by_header[tier].begin(), db.patches_by_header[tier].end(),
			[](const tier_patch& lhs, const tier_patch& rhs) { return lhs.segment_header_offset < rhs.segment_header_offset; });
		// A segment's keyframes of one tier sit in ONE chunk (build_database writes a segment's tier whole): two chunk segment headers
		// that name the same runtime segment header would have the later stream-in overwrite the earlier one's tier words in the
		// reference (database.impl.h:520-535, chunk order) and race here, where the patches of a request are applied in parallel.
		// Refused. (Found by tools/fuzz_gpu_mutated_db.py, round 6: a segment_header_offset moved onto its neighbour's.)
		for (size_t i = 1; i < db.patches_by_header[tier].size(); ++i)
			if (db.patches_by_header[tier][i].segment_header_offset == db.patches_by_header[tier][i - 1].segment_header_offset)
				return fail(context, ACLHIP_ERROR_INVALID_CLIP, "Tier %d: two chunks hold keyframes of the same segment", tier + 1);
	}
	if (validate_only)
		return ACLHIP_OK;		// aclhip_check_database: everything above is host work

	std::lock_guard<std::shared_mutex> lock(context->mutex);
	device_guard guard(context->device);
	if (!guard.ok)
		return fail(context, ACLHIP_ERROR_DEVICE, "hipSetDevice(%d) failed", context->device);

	// (uploads on the context's copy stream: registering a database does not stall decodes in flight)
	collect_retired(context, false);
	size_t staging_used = 0;
	bool ok = hipMalloc(reinterpret_cast<void**>(&db.d_runtime_headers), runtime.size()) == hipSuccess
		&& stage_upload(context, db.d_runtime_headers, runtime.data(), runtime.size(), staging_used);
	for (int tier = 0; tier < 2 && ok; ++tier)
	{
		// +64: keyframe windows of up to 16 bytes are read past the last sample, the reference reserves 15 (compress.database.impl.h:910)
		const size_t bulk_bytes = size_t(header.bulk_data_size[tier]) + 64;
		ok = hipMalloc(reinterpret_cast<void**>(&db.d_bulk_data[tier]), bulk_bytes) == hipSuccess
			&& hipMemsetAsync(db.d_bulk_data[tier], 0xCD, bulk_bytes, context->copy_stream) == hipSuccess		// like debug_database_streamer: not-resident memory is poison
			&& hipMalloc(reinterpret_cast<void**>(&db.d_patches[tier]), std::max<size_t>(streamed ? header.num_segments : patches[tier].size(), 1) * sizeof(tier_patch)) == hipSuccess;
		db.patch_capacity[tier] = uint32_t(streamed ? header.num_segments : patches[tier].size());		// a segment has at most one entry per tier
		if (ok && !patches[tier].empty())
			ok = stage_upload(context, db.d_patches[tier], patches[tier].data(), patches[tier].size() * sizeof(tier_patch), staging_used);
		if (ok && streamed && header.num_segments != 0 && num_descriptions != 0)
			ok = hipHostMalloc(reinterpret_cast<void**>(&db.pinned_patches[tier]), size_t(header.num_segments) * sizeof(tier_patch), hipHostMallocDefault) == hipSuccess;
		if (ok && header.bulk_data_size[tier] != 0)
		{
			ok = hipHostMalloc(reinterpret_cast<void**>(&db.pinned_bulk_data[tier]), header.bulk_data_size[tier], hipHostMallocDefault) == hipSuccess;
			if (ok && !streamed)
				std::memcpy(db.pinned_bulk_data[tier], bulk_sources[tier], header.bulk_data_size[tier]);
		}
	}
	ok = ok && finish_uploads(context);
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

extern "C" aclhip_status aclhip_register_database_streamed(aclhip_context* context, const void* compressed_database, uint64_t size, int check_hash, aclhip_database* out_database)
{
	return guarded(context, [&]() { return register_database_impl(context, compressed_database, size, nullptr, nullptr, check_hash, out_database, false, true); });
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

extern "C" aclhip_status aclhip_analyze_clip(const void* compressed_tracks, uint64_t size, int check_hash, uint32_t* out_facts)
{
	if (out_facts == nullptr)
		return ACLHIP_ERROR_INVALID_ARGUMENT;
	aclhip_context scratch;		// no device is touched
	aclhip_clip unused = ACLHIP_INVALID_HANDLE;
	return guarded(&scratch, [&]() { return register_clip_impl(&scratch, compressed_tracks, size, check_hash, ACLHIP_INVALID_HANDLE, &unused, true, out_facts); });
}

extern "C" aclhip_status aclhip_check_database(const void* compressed_database, uint64_t size, const void* bulk_data_medium, const void* bulk_data_low,
	int check_hash, char* out_message, uint32_t capacity)
{
	aclhip_context scratch;
	aclhip_database unused = ACLHIP_INVALID_HANDLE;
	return report(scratch, guarded(&scratch, [&]() { return register_database_impl(&scratch, compressed_database, size, bulk_data_medium, bulk_data_low, check_hash, &unused, true); }), out_message, capacity);
}

// strip_database_quality_tier (compression/impl/compress.database.impl.h:1388-1525), host only: the compressed_database without one
// of its two streamable tiers, byte for byte what the reference builds -- same layout arithmetic, chunk descriptions and clip
// metadata copied, the stripped tier's counts zeroed and its hash set to hash32(nullptr, 0), buffer hash recomputed. Like the
// reference, room for the remaining tier's bulk data is reserved (and its offset set) whether or not the bulk data is inline, and
// only inline bulk data is copied.
extern "C" aclhip_status aclhip_strip_database_tier(const void* compressed_database, uint64_t size, uint32_t tier, void* out_database, uint64_t capacity, uint64_t* out_size)
{
	if (out_size == nullptr)
		return ACLHIP_ERROR_INVALID_ARGUMENT;
	*out_size = 0;

	// database.is_valid(true) (core/impl/compressed_database.impl.h:142-163)
	const uint8_t* blob = static_cast<const uint8_t*>(compressed_database);
	if (blob == nullptr || size < sizeof(raw_buffer_header) + sizeof(database_header) || (reinterpret_cast<uintptr_t>(blob) & 7u) != 0)
		return ACLHIP_ERROR_INVALID_CLIP;
	const raw_buffer_header& buffer_header = *reinterpret_cast<const raw_buffer_header*>(blob);
	const database_header& header = *reinterpret_cast<const database_header*>(blob + sizeof(raw_buffer_header));
	const uint8_t* hbase = reinterpret_cast<const uint8_t*>(&header);
	if (header.tag != k_tag_compressed_database || header.version < k_version_first || header.version > k_version_latest
		|| buffer_header.size > size || buffer_header.size < sizeof(raw_buffer_header) + sizeof(database_header)
		|| hash32(blob + sizeof(raw_buffer_header), buffer_header.size - sizeof(raw_buffer_header)) != buffer_header.hash)
		return ACLHIP_ERROR_INVALID_CLIP;
	const uint64_t header_limit = buffer_header.size - sizeof(raw_buffer_header);
	const uint32_t descriptions_offset = align_to_u32(uint32_t(sizeof(database_header)), 4);
	if (uint64_t(descriptions_offset) + (uint64_t(header.num_chunks[0]) + header.num_chunks[1]) * sizeof(database_chunk_description) > header_limit
		|| uint64_t(header.clip_metadata_offset) + uint64_t(header.num_clips) * sizeof(database_clip_metadata) > header_limit)
		return ACLHIP_ERROR_INVALID_CLIP;

	// the high importance tier lives inside the compressed_tracks; an empty tier cannot be stripped (:1396-1400)
	if (tier != 1 && tier != 2)
		return ACLHIP_ERROR_INVALID_ARGUMENT;
	const uint32_t tier_index = tier - 1;
	if (header.bulk_data_size[tier_index] == 0)
		return ACLHIP_ERROR_INVALID_ARGUMENT;

	const bool is_inline = (header.misc_packed & 1u) != 0;
	const uint32_t kept_index = 1 - tier_index;
	const uint32_t num_kept_chunks = header.num_chunks[kept_index];
	const uint32_t kept_bulk_size = header.bulk_data_size[kept_index];
	if (is_inline && kept_bulk_size != 0 && (header.bulk_data_offset[kept_index] == k_invalid_offset || uint64_t(header.bulk_data_offset[kept_index]) + kept_bulk_size > header_limit))
		return ACLHIP_ERROR_INVALID_CLIP;

	// offsets relative to the database_header, in the reference's order: chunk descriptions (medium, low), clip metadata, bulk data
	constexpr uint32_t k_bulk_data_alignment = 4;		// alignof(database_chunk_header) (core/compressed_database.h:170)
	uint64_t offset = descriptions_offset;
	const uint64_t kept_descriptions_offset = offset;		// the stripped tier has none: the kept tier's come first either way
	offset += uint64_t(num_kept_chunks) * sizeof(database_chunk_description);
	offset = (offset + 3) & ~uint64_t(3);
	const uint64_t clip_metadata_offset = offset;
	offset += uint64_t(header.num_clips) * sizeof(database_clip_metadata);
	offset = (offset + sizeof(raw_buffer_header) + k_bulk_data_alignment - 1) / k_bulk_data_alignment * k_bulk_data_alignment - sizeof(raw_buffer_header);
	const uint64_t bulk_data_offset = offset;
	offset += kept_bulk_size;
	const uint64_t total_size = sizeof(raw_buffer_header) + offset;
	if (total_size > 0xFFFFFFFFull)
		return ACLHIP_ERROR_INVALID_CLIP;

	*out_size = total_size;
	if (out_database == nullptr || capacity < total_size)
		return out_database == nullptr && capacity == 0 ? ACLHIP_OK : ACLHIP_ERROR_INVALID_ARGUMENT;		// size query

	uint8_t* out = static_cast<uint8_t*>(out_database);
	std::memset(out, 0, total_size);
	raw_buffer_header* out_buffer_header = reinterpret_cast<raw_buffer_header*>(out);
	database_header* out_header = reinterpret_cast<database_header*>(out + sizeof(raw_buffer_header));
	uint8_t* out_hbase = reinterpret_cast<uint8_t*>(out_header);
	std::memcpy(out_header, &header, sizeof(database_header));
	out_header->clip_metadata_offset = uint32_t(clip_metadata_offset);
	out_header->bulk_data_offset[kept_index] = kept_bulk_size != 0 ? uint32_t(bulk_data_offset) : k_invalid_offset;
	out_header->num_chunks[tier_index] = 0;
	out_header->bulk_data_size[tier_index] = 0;
	out_header->bulk_data_offset[tier_index] = k_invalid_offset;
	out_header->bulk_data_hash[tier_index] = hash32(nullptr, 0);

	const uint64_t source_descriptions_offset = kept_index == 0 ? descriptions_offset
		: ((uint64_t(descriptions_offset) + uint64_t(header.num_chunks[0]) * sizeof(database_chunk_description) + 3) & ~uint64_t(3));
	std::memcpy(out_hbase + kept_descriptions_offset, hbase + source_descriptions_offset, size_t(num_kept_chunks) * sizeof(database_chunk_description));
	std::memcpy(out_hbase + clip_metadata_offset, hbase + header.clip_metadata_offset, size_t(header.num_clips) * sizeof(database_clip_metadata));
	if (is_inline && kept_bulk_size != 0)
		std::memcpy(out_hbase + bulk_data_offset, hbase + header.bulk_data_offset[kept_index], kept_bulk_size);

	out_buffer_header->size = uint32_t(total_size);
	out_buffer_header->hash = hash32(out_hbase, total_size - sizeof(raw_buffer_header));
	return ACLHIP_OK;
}

extern "C" aclhip_status aclhip_unregister_database(aclhip_context* context, aclhip_database database)
{
	if (context == nullptr)
		return ACLHIP_ERROR_INVALID_ARGUMENT;
	std::lock_guard<std::shared_mutex> lock(context->mutex);
	if (database >= context->databases.size() || !context->databases[database].in_use)
		return fail(context, ACLHIP_ERROR_UNKNOWN_DATABASE, "unknown database handle %u", database);
	if (context->databases[database].num_bound_clips != 0)
		return fail(context, ACLHIP_ERROR_INVALID_ARGUMENT, "%u clips are still bound to this database", context->databases[database].num_bound_clips);
	device_guard guard(context->device);
	collect_retired(context, false);
	// retired, not freed: stream-in copies and decodes already enqueued may still touch the tiers (freed once they have completed)
	host_database& db = context->databases[database];
	aclhip_context::retired_item item;
	item.database_memory[0] = db.d_runtime_headers;
	for (int tier = 0; tier < 2; ++tier)
	{
		item.database_memory[1 + tier] = db.d_bulk_data[tier];
		item.database_memory[3 + tier] = reinterpret_cast<uint8_t*>(db.d_patches[tier]);
		item.database_pinned[tier] = db.pinned_bulk_data[tier];
		item.database_pinned[2 + tier] = reinterpret_cast<uint8_t*>(db.pinned_patches[tier]);
	}
	item.database_memory[5] = reinterpret_cast<uint8_t*>(db.d_bound_clips);
	retire(context, std::move(item));
	db = host_database();
	return ACLHIP_OK;
}

extern "C" aclhip_status aclhip_get_database_info(const aclhip_context* context, aclhip_database database, aclhip_database_info* out_info)
{
	if (context == nullptr || out_info == nullptr)
		return ACLHIP_ERROR_INVALID_ARGUMENT;
	std::lock_guard<std::shared_mutex> lock(const_cast<aclhip_context*>(context)->mutex);
	if (database >= context->databases.size() || !context->databases[database].in_use)
		return fail(context, ACLHIP_ERROR_UNKNOWN_DATABASE, "unknown database handle %u", database);
	*out_info = context->databases[database].info;
	return ACLHIP_OK;
}

namespace
{
	// Streamed databases: chunks [first, last] of `tier_bulk_data` (the tier's bulk data as the caller's streamer holds it) have just
	// arrived. Chunks seen for the first time are checked and turned into metadata patches -- what registration does up front when it is
	// given the bulk data (stream_in only ever continues behind the last resident chunk: first arrivals come in chunk order) -- and the
	// bytes are taken over into the pinned backing store the device copy is made from.
	aclhip_status take_arrived_chunks(aclhip_context* context, host_database& db, uint32_t tier_index, uint32_t first_chunk_index, uint32_t last_chunk_index,
		const uint8_t* tier_bulk_data, hipStream_t hip_stream)
	{
		// Untrusted input, two phases: every chunk of the request is checked and its patches collected BEFORE anything of the database's
		// state changes -- a request that fails half way leaves nothing parsed-but-not-uploaded behind (the kernel that applies the
		// patches would read table entries nobody wrote) --, then the state is committed and the new patches are uploaded; what a failed
		// copy left behind is uploaded by the next call (num_uploaded_patches).
		const uint32_t bulk_size = db.info.bulk_data_size[tier_index];
		uint32_t next_chunk = db.num_parsed_chunks[tier_index];
		uint32_t next_patch = db.chunk_first_patch[tier_index][next_chunk];
		std::vector<tier_patch> new_patches;
		std::unordered_set<uint32_t> new_headers;	// segment headers the new patches name
		std::vector<uint32_t> new_chunk_ends;		// patches up to and including every newly parsed chunk
		for (uint32_t chunk_index = first_chunk_index; chunk_index <= last_chunk_index; ++chunk_index)
		{
			const database_chunk_description& description = db.chunks[tier_index][chunk_index];
			const database_chunk_header& chunk = *reinterpret_cast<const database_chunk_header*>(tier_bulk_data + description.offset);
			if (chunk.index != chunk_index || chunk.size != description.size
				|| uint64_t(sizeof(database_chunk_header)) + uint64_t(chunk.num_segments) * sizeof(database_chunk_segment_header) > description.size)
				return fail(context, ACLHIP_ERROR_INVALID_CLIP, "Chunk %u of tier %u arrived with an invalid header", chunk_index, tier_index + 1);
			if (chunk_index < next_chunk)
				continue;		// seen before (streamed out and back in): its patches are known
			if (chunk_index != next_chunk)
				return fail(context, ACLHIP_ERROR_INVALID_ARGUMENT, "chunk %u of tier %u arrived before chunk %u", chunk_index, tier_index + 1, next_chunk);

			const database_chunk_segment_header* segments = reinterpret_cast<const database_chunk_segment_header*>(&chunk + 1);
			if (uint64_t(next_patch) + chunk.num_segments > db.patch_capacity[tier_index])
				return fail(context, ACLHIP_ERROR_INVALID_CLIP, "Chunk %u of tier %u lists more segments than the database has", chunk_index, tier_index + 1);
			for (uint32_t i = 0; i < chunk.num_segments; ++i)
			{
				if (uint64_t(segments[i].segment_header_offset) + sizeof(database_runtime_segment_header) > db.runtime_headers_size || segments[i].samples_offset >= bulk_size
					|| (segments[i].segment_header_offset & 7u) != 0 || !names_a_segment_header(db, segments[i].clip_header_offset, segments[i].segment_header_offset))
					return fail(context, ACLHIP_ERROR_INVALID_CLIP, "Chunk %u of tier %u points outside of the database", chunk_index, tier_index + 1);
				// keyframes of a clip that is already bound must lie inside the tier (clips bound later are checked when they are bound)
				for (const std::pair<uint32_t, uint32_t>& bound : db.segment_pose_bits)
					if (bound.first == segments[i].segment_header_offset
						&& uint64_t(segments[i].samples_offset) + (uint64_t(__builtin_popcount(segments[i].sample_indices)) * bound.second + 7) / 8 > bulk_size)
						return fail(context, ACLHIP_ERROR_INVALID_CLIP, "Chunk %u of tier %u: keyframes lie outside of the bulk data", chunk_index, tier_index + 1);
				// ... and a segment has its keyframes of this tier in one chunk (see register_database_impl)
				const std::vector<tier_patch>& known = db.patches_by_header[tier_index];
				const auto same_header = std::lower_bound(known.begin(), known.end(), segments[i].segment_header_offset,
					[](const tier_patch& patch, uint32_t offset) { return patch.segment_header_offset < offset; });
				bool repeated = same_header != known.end() && same_header->segment_header_offset == segments[i].segment_header_offset;
				repeated = repeated || !new_headers.insert(segments[i].segment_header_offset).second;
				if (repeated)
					return fail(context, ACLHIP_ERROR_INVALID_CLIP, "Chunk %u of tier %u: another chunk holds keyframes of the same segment", chunk_index, tier_index + 1);
				new_patches.push_back({ segments[i].segment_header_offset, segments[i].sample_indices, segments[i].samples_offset });
			}
			next_patch += chunk.num_segments;
			next_chunk = chunk_index + 1;
			new_chunk_ends.push_back(next_patch);
		}

		// commit: bytes into the backing store, patches into the mirror and the lookup, chunk bookkeeping
		for (uint32_t chunk_index = first_chunk_index; chunk_index <= last_chunk_index; ++chunk_index)
		{
			const database_chunk_description& description = db.chunks[tier_index][chunk_index];
			std::memcpy(db.pinned_bulk_data[tier_index] + description.offset, tier_bulk_data + description.offset, description.size);
		}
		const uint32_t first_new_patch = db.chunk_first_patch[tier_index][db.num_parsed_chunks[tier_index]];
		for (size_t i = 0; i < new_patches.size(); ++i)
		{
			const tier_patch& patch = new_patches[i];
			db.pinned_patches[tier_index][first_new_patch + i] = patch;
			db.patches_by_header[tier_index].insert(std::upper_bound(db.patches_by_header[tier_index].begin(), db.patches_by_header[tier_index].end(), patch,
				[](const tier_patch& lhs, const tier_patch& rhs) { return lhs.segment_header_offset < rhs.segment_header_offset; }), patch);
		}
		for (uint32_t end : new_chunk_ends)
			db.chunk_first_patch[tier_index][++db.num_parsed_chunks[tier_index]] = end;

		// upload whatever of the parsed chunks' patches is not on the device yet (this request's, and what an earlier failed copy left)
		const uint32_t parsed_patches = db.chunk_first_patch[tier_index][db.num_parsed_chunks[tier_index]];
		if (parsed_patches != db.num_uploaded_patches[tier_index])
		{
			const uint32_t first_pending = db.num_uploaded_patches[tier_index];
			ACLHIP_CHECK_HIP(context, hipMemcpyAsync(db.d_patches[tier_index] + first_pending, db.pinned_patches[tier_index] + first_pending,
				size_t(parsed_patches - first_pending) * sizeof(tier_patch), hipMemcpyHostToDevice, hip_stream));
			db.num_uploaded_patches[tier_index] = parsed_patches;
		}
		return ACLHIP_OK;
	}

	aclhip_status stream_database(aclhip_context* context, aclhip_database database, uint32_t tier, uint32_t num_chunks_to_stream, void* stream, bool stream_in, uint32_t* out_num_chunks,
		const void* tier_bulk_data = nullptr)
	{
		if (context == nullptr)
			return ACLHIP_ERROR_INVALID_ARGUMENT;
		if (out_num_chunks != nullptr)
			*out_num_chunks = 0;
		std::lock_guard<std::shared_mutex> lock(context->mutex);
		if (database >= context->databases.size() || !context->databases[database].in_use)
			return fail(context, ACLHIP_ERROR_UNKNOWN_DATABASE, "unknown database handle %u", database);
		if (tier != 1 && tier != 2)
			return fail(context, ACLHIP_ERROR_INVALID_ARGUMENT, "tier must be 1 (medium importance) or 2 (lowest importance)");	// invalid_database_tier

		host_database& db = context->databases[database];
		const uint32_t tier_index = tier - 1;
		const uint32_t num_chunks = db.info.num_chunks[tier_index];
		if (stream_in && db.streamed != (tier_bulk_data != nullptr) && num_chunks != 0)
			return fail(context, ACLHIP_ERROR_INVALID_ARGUMENT, db.streamed ? "a streamed database takes its chunks through aclhip_database_stream_in_from" : "this database was registered with its bulk data");
		num_chunks_to_stream = std::min(num_chunks_to_stream, num_chunks);
		if (num_chunks == 0)
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

		// The reference's own arithmetic (database.impl.h:490-497,571-578), quirk included: a request for 0 chunks computes
		// first + 0 - 1, which wraps when the first candidate is chunk 0 -- the WHOLE tier moves -- and is first - 1 otherwise: nothing moves
		const uint64_t last_chunk_index64 = uint64_t(first_chunk_index) + uint64_t(num_chunks_to_stream) - 1;
		const uint32_t last_chunk_index = last_chunk_index64 >= uint64_t(num_chunks) ? num_chunks - 1 : uint32_t(last_chunk_index64);
		const uint32_t num_streaming_chunks = last_chunk_index - first_chunk_index + 1;
		if (num_streaming_chunks == 0)
			return ACLHIP_OK;	// database_stream_request_result::done

		device_guard guard(context->device);
		hipStream_t hip_stream = static_cast<hipStream_t>(stream);
		note_launch_stream(context, hip_stream);
		if (stream_in && db.streamed)
		{
			const aclhip_status arrived = take_arrived_chunks(context, db, tier_index, first_chunk_index, last_chunk_index, static_cast<const uint8_t*>(tier_bulk_data), hip_stream);
			if (arrived != ACLHIP_OK)
				return arrived;
		}
		// (a streamed database only knows the chunks that have arrived; none beyond them can be resident)
		const uint32_t known_end = std::min(last_chunk_index + 1, db.num_parsed_chunks[tier_index]);
		const uint32_t first_patch = db.chunk_first_patch[tier_index][std::min(first_chunk_index, known_end)];
		const uint32_t num_patches = db.chunk_first_patch[tier_index][known_end] - first_patch;

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
			// ... and into the sample records of the clips bound to the database, where the decode reads them (database_sample_record)
			if (!db.bound_clips.empty())
			{
				hipLaunchKernelGGL(refresh_database_sample_tiers_kernel, dim3(uint32_t(db.bound_clips.size())), dim3(256), 0, hip_stream,
					context->d_clips, context->d_clips_capacity, db.d_bound_clips, uint32_t(db.bound_clips.size()), db.d_runtime_headers);
				ACLHIP_CHECK_HIP(context, hipGetLastError());
			}
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

extern "C" aclhip_status aclhip_database_stream_in_from(aclhip_context* context, aclhip_database database, uint32_t tier, uint32_t num_chunks, const void* tier_bulk_data,
	void* stream, uint32_t* out_num_chunks)
{
	if (tier_bulk_data == nullptr)
		return context != nullptr ? fail(context, ACLHIP_ERROR_INVALID_ARGUMENT, "null bulk data") : ACLHIP_ERROR_INVALID_ARGUMENT;
	return stream_database(context, database, tier, num_chunks, stream, true, out_num_chunks, tier_bulk_data);
}

extern "C" aclhip_status aclhip_database_stream_out(aclhip_context* context, aclhip_database database, uint32_t tier, uint32_t num_chunks, void* stream, uint32_t* out_num_chunks)
{
	return stream_database(context, database, tier, num_chunks, stream, false, out_num_chunks);
}

extern "C" aclhip_status aclhip_get_clip_info(const aclhip_context* context, aclhip_clip clip, aclhip_clip_info* out_info)
{
	if (context == nullptr || out_info == nullptr)
		return ACLHIP_ERROR_INVALID_ARGUMENT;
	std::lock_guard<std::shared_mutex> lock(const_cast<aclhip_context*>(context)->mutex);
	if (clip >= context->clips.size() || !context->clips[clip].in_use)
		return fail(context, ACLHIP_ERROR_UNKNOWN_CLIP, "unknown clip handle %u", clip);
	*out_info = context->clips[clip].info;
	return ACLHIP_OK;
}

// Host only: compressed_tracks::get_parent_track_index / get_track_description on a blob that need not be registered
extern "C" aclhip_status aclhip_read_clip_metadata(const void* compressed_tracks, uint64_t size, aclhip_clip_metadata_info* out_info, uint32_t* out_parent_indices,
	float* out_default_values, float* out_precisions, float* out_shell_distances, uint32_t capacity)
{
	if (compressed_tracks == nullptr || out_info == nullptr)
		return ACLHIP_ERROR_INVALID_ARGUMENT;
	try
	{
		const uint8_t* blob = static_cast<const uint8_t*>(compressed_tracks);
		const aclhip_status status = validate_clip(nullptr, blob, size, 0);
		if (status != ACLHIP_OK)
			return status;
		const raw_buffer_header& buffer_header = *reinterpret_cast<const raw_buffer_header*>(blob);
		const tracks_header& header = *reinterpret_cast<const tracks_header*>(blob + k_tracks_header_offset);
		std::vector<uint32_t> parents;
		std::vector<float> descriptions;
		parse_clip_metadata(blob, buffer_header.size, header, *out_info, parents, descriptions);
		if ((out_parent_indices != nullptr && out_info->has_parent_track_indices != 0) || ((out_default_values != nullptr || out_precisions != nullptr || out_shell_distances != nullptr) && out_info->has_track_descriptions != 0))
			if (capacity < header.num_tracks)
				return fail(nullptr, ACLHIP_ERROR_INVALID_ARGUMENT, "room for %u tracks, the clip has %u", capacity, header.num_tracks);
		if (out_parent_indices != nullptr && !parents.empty())
			std::memcpy(out_parent_indices, parents.data(), parents.size() * sizeof(uint32_t));
		for (size_t track = 0; track < descriptions.size() / 14; ++track)
		{
			const float* row = &descriptions[track * 14];
			if (out_default_values != nullptr)
				std::memcpy(out_default_values + track * 12, row, 48);
			if (out_precisions != nullptr)
				out_precisions[track] = row[12];
			if (out_shell_distances != nullptr)
				out_shell_distances[track] = row[13];
		}
		return ACLHIP_OK;
	}
	catch (const std::bad_alloc&)
	{
		return ACLHIP_ERROR_OUT_OF_MEMORY;
	}
}

extern "C" aclhip_status aclhip_get_clip_metadata_info(const aclhip_context* context, aclhip_clip clip, aclhip_clip_metadata_info* out_info)
{
	if (context == nullptr || out_info == nullptr)
		return ACLHIP_ERROR_INVALID_ARGUMENT;
	std::lock_guard<std::shared_mutex> lock(const_cast<aclhip_context*>(context)->mutex);
	if (clip >= context->clips.size() || !context->clips[clip].in_use)
		return fail(context, ACLHIP_ERROR_UNKNOWN_CLIP, "unknown clip handle %u", clip);
	*out_info = context->clips[clip].metadata;
	return ACLHIP_OK;
}

// compressed_tracks::get_parent_track_index (core/impl/compressed_tracks.impl.h:175-190) for every track
extern "C" aclhip_status aclhip_get_clip_parent_indices(const aclhip_context* context, aclhip_clip clip, uint32_t* out_parent_indices, uint32_t capacity)
{
	if (context == nullptr || (out_parent_indices == nullptr && capacity != 0))
		return ACLHIP_ERROR_INVALID_ARGUMENT;
	std::lock_guard<std::shared_mutex> lock(const_cast<aclhip_context*>(context)->mutex);
	if (clip >= context->clips.size() || !context->clips[clip].in_use)
		return fail(context, ACLHIP_ERROR_UNKNOWN_CLIP, "unknown clip handle %u", clip);
	const host_clip& entry = context->clips[clip];
	if (entry.metadata.has_parent_track_indices == 0)
		return fail(context, ACLHIP_ERROR_NO_METADATA, "clip %u was compressed without include_parent_track_indices", clip);
	if (capacity < entry.metadata_parents.size())
		return fail(context, ACLHIP_ERROR_INVALID_ARGUMENT, "room for %u parent indices, the clip has %zu tracks", capacity, entry.metadata_parents.size());
	if (!entry.metadata_parents.empty())
		std::memcpy(out_parent_indices, entry.metadata_parents.data(), entry.metadata_parents.size() * sizeof(uint32_t));
	return ACLHIP_OK;
}

// compressed_tracks::get_track_description(track, track_desc_transformf&) (core/impl/compressed_tracks.impl.h:214-275) for every track
extern "C" aclhip_status aclhip_get_clip_track_descriptions(const aclhip_context* context, aclhip_clip clip, float* out_default_values, float* out_precisions, float* out_shell_distances, uint32_t capacity)
{
	if (context == nullptr)
		return ACLHIP_ERROR_INVALID_ARGUMENT;
	std::lock_guard<std::shared_mutex> lock(const_cast<aclhip_context*>(context)->mutex);
	if (clip >= context->clips.size() || !context->clips[clip].in_use)
		return fail(context, ACLHIP_ERROR_UNKNOWN_CLIP, "unknown clip handle %u", clip);
	const host_clip& entry = context->clips[clip];
	if (entry.metadata.has_track_descriptions == 0)
		return fail(context, ACLHIP_ERROR_NO_METADATA, "clip %u was compressed without include_track_descriptions", clip);
	const size_t num_tracks = entry.metadata_descriptions.size() / 14;
	if (capacity < num_tracks)
		return fail(context, ACLHIP_ERROR_INVALID_ARGUMENT, "room for %u track descriptions, the clip has %zu tracks", capacity, num_tracks);
	for (size_t track = 0; track < num_tracks; ++track)
	{
		const float* row = &entry.metadata_descriptions[track * 14];
		if (out_default_values != nullptr)
			std::memcpy(out_default_values + track * 12, row, 48);
		if (out_precisions != nullptr)
			out_precisions[track] = row[12];
		if (out_shell_distances != nullptr)
			out_shell_distances[track] = row[13];
	}
	return ACLHIP_OK;
}

extern "C" aclhip_status aclhip_clip_matches(const aclhip_context* context, aclhip_clip clip, const void* compressed_tracks, int* out_matches)
{
	if (context == nullptr || compressed_tracks == nullptr || out_matches == nullptr)
		return ACLHIP_ERROR_INVALID_ARGUMENT;
	std::lock_guard<std::shared_mutex> lock(const_cast<aclhip_context*>(context)->mutex);
	if (clip >= context->clips.size() || !context->clips[clip].in_use)
		return fail(context, ACLHIP_ERROR_UNKNOWN_CLIP, "unknown clip handle %u", clip);
	// is_bound_to_v0 compares pointer and hash (decompression.transform.h:159-169); there is no shared pointer here: hash + size
	const raw_buffer_header& buffer_header = *static_cast<const raw_buffer_header*>(compressed_tracks);
	const aclhip_clip_info& info = context->clips[clip].info;
	*out_matches = (buffer_header.hash == info.hash && buffer_header.size == info.compressed_size) ? 1 : 0;
	return ACLHIP_OK;
}
