/* The call sequences INTEGRATION.md shows (sections 1 incl. instance lists and per track skips, 3, 4b, 5), as one C99 translation unit that must compile and link against
 * libaclhip.so with the signatures the document uses. Never run: tests/test_capi_symbols.py only builds it. */
#include "aclhip.h"

#include <stddef.h>

int integration_example(const void* tracks, uint64_t tracks_size, const void* compressed_db, uint64_t db_size, const void* bulk_medium, const void* bulk_low,
	const aclhip_clip* d_clips, const float* d_times, const uint32_t* d_bones, uint32_t num_instances, void* d_poses, void* d_transforms, uint32_t max_tracks,
	const uint32_t* parent_indices, uint32_t num_tracks, const aclhip_clip* d_base_clips, const float* d_base_times,
	const aclhip_clip* host_clips, uint32_t* order, uint32_t* d_rows, aclhip_clip* d_ordered_clips, float* d_ordered_times, void* hip_stream)
{
	aclhip_context* gpu = NULL;
	aclhip_clip clip, db_clip;
	aclhip_database db;
	aclhip_decompress_params params;
	aclhip_pose_consumers consumers;
	char message[256];
	uint32_t moved = 0;
	aclhip_status s;

	/* section 1 */
	if (aclhip_create(0, &gpu) != ACLHIP_OK)
		return 1;
	s = aclhip_register_clip(gpu, tracks, tracks_size, 0, &clip);
	if (s == ACLHIP_ERROR_INVALID_CLIP)
		(void)aclhip_check_clip(tracks, tracks_size, 0, message, sizeof(message));
	aclhip_default_params(&params);
	params.rounding_policy = ACLHIP_ROUND_NONE;
	s = aclhip_decompress_tracks_batch(gpu, d_clips, d_times, num_instances, &params, d_poses, (uint64_t)max_tracks * 48, hip_stream);
	s = aclhip_decompress_track_batch(gpu, d_clips, d_times, d_bones, num_instances, &params, d_transforms, hip_stream);
	s = aclhip_order_instances_for_locality(gpu, host_clips, num_instances, order);
	s = aclhip_order_instances_for_pose_windows(1, host_clips, num_instances, order);
	s = aclhip_order_instances_device(gpu, d_clips, d_times, num_instances, d_rows, d_ordered_clips, d_ordered_times, hip_stream);
	s = aclhip_decompress_tracks_batch(gpu, d_ordered_clips, d_ordered_times, num_instances, NULL, d_poses, (uint64_t)max_tracks * 48, hip_stream);
	s = aclhip_decompress_tracks_batch_rows(gpu, d_clips, d_times, d_rows, num_instances, &params, d_poses, (uint64_t)max_tracks * 48, hip_stream);
	{
		aclhip_output_desc output = { 0 };
		output.layout = ACLHIP_LAYOUT_QV32;
		s = aclhip_decompress_tracks_batch_out(gpu, d_clips, d_times, num_instances, &params, &output, d_poses, (uint64_t)max_tracks * 32, hip_stream);
	}
	{
		/* an instance list that keeps its decode order across frames; per track skips; a stream the engine destroys */
		aclhip_instance_list characters;
		const uint32_t* d_order = NULL;
		uint64_t orderings = 0;
		aclhip_output_desc output = { 0 };
		output.skip_tracks = (const uint8_t*)d_bones;		/* (one byte per track in device memory) */
		s = aclhip_instance_list_create(gpu, num_instances, &characters);
		s = aclhip_instance_list_set_clips(gpu, characters, d_clips, hip_stream);
		s = aclhip_instance_list_update(gpu, characters, d_rows, d_ordered_clips, 3, hip_stream);
		s = aclhip_decompress_tracks_list(gpu, characters, d_times, NULL, NULL, 0, d_poses, (uint64_t)max_tracks * 48, hip_stream);
		s = aclhip_decompress_tracks_list(gpu, characters, d_times, &params, &output, 1, d_poses, (uint64_t)max_tracks * 48, hip_stream);
		s = aclhip_instance_list_get_order(gpu, characters, &d_order, &orderings);
		{
			/* per character writers and contexts (ABI 5): LODs, skip masks, looping policies and per track rounding tables per instance; a list
			 * attached to the engine's own clip array */
			aclhip_output_desc writers = { 0 };
			aclhip_decompress_params per_character = params;
			writers.instance_track_counts = d_rows;
			writers.mask_table = (const uint8_t*)d_bones;
			writers.mask_stride = max_tracks;
			writers.instance_masks = (const uint8_t*)d_bones;
			per_character.instance_looping_policies = (const uint8_t*)d_bones;
			per_character.per_track_rounding = 1;
			per_character.rounding_policy = ACLHIP_ROUND_PER_TRACK;
			per_character.track_rounding_table = (const uint8_t*)d_bones;
			per_character.track_rounding_stride = max_tracks;
			per_character.instance_rounding_tables = (const uint8_t*)d_bones;
			s = aclhip_decompress_tracks_batch_out(gpu, d_clips, d_times, num_instances, &per_character, &writers, d_poses, (uint64_t)max_tracks * 48, hip_stream);
			s = aclhip_instance_list_attach(gpu, characters, d_clips, hip_stream);
			s = aclhip_instance_list_note_changes(gpu, characters, 3);
			s = aclhip_decompress_tracks_list(gpu, characters, d_times, &per_character, &writers, 0, d_poses, (uint64_t)max_tracks * 48, hip_stream);
		}
		{
			int rccl_version = 0;
			char rccl_path[256], rccl_how[64];
			s = aclhip_probe_rccl(&rccl_version, rccl_path, sizeof(rccl_path), rccl_how, sizeof(rccl_how));
		}
		s = aclhip_instance_list_destroy(gpu, characters);
		s = aclhip_forget_stream(gpu, hip_stream);
		if (aclhip_abi_version() != ACLHIP_ABI_VERSION)
			return 1;
	}

	/* section 3 */
	s = aclhip_register_database(gpu, compressed_db, db_size, bulk_medium, bulk_low, 0, &db);
	s = aclhip_register_clip_with_database(gpu, tracks, tracks_size, 0, db, &db_clip);
	s = aclhip_database_stream_in(gpu, db, 1, 4, hip_stream, &moved);
	s = aclhip_database_stream_out(gpu, db, 2, ~0u, hip_stream, &moved);
	{
		/* the engine's own streamers serve the bulk data */
		aclhip_database streamed_db;
		s = aclhip_register_database_streamed(gpu, compressed_db, db_size, 0, &streamed_db);
		s = aclhip_database_stream_in_from(gpu, streamed_db, 1, 2, bulk_medium, hip_stream, &moved);
		(void)aclhip_unregister_database(gpu, streamed_db);
	}

	/* section 4b */
	s = aclhip_set_clip_hierarchy(gpu, clip, parent_indices, num_tracks);
	consumers.additive_format = ACLHIP_ADDITIVE_ADDITIVE1;
	consumers.object_space = 1;
	consumers.base_clips = d_base_clips;
	consumers.base_sample_times = d_base_times;
	consumers.base_poses = NULL;
	consumers.base_pose_stride_bytes = 0;
	s = aclhip_decompress_poses_batch(gpu, d_clips, d_times, num_instances, &params, &consumers, d_poses, (uint64_t)max_tracks * 48, hip_stream);

	/* section 5 */
	{
		uint8_t handle[ACLHIP_PEER_HANDLE_BYTES];
		void* peer = NULL;
		s = aclhip_peer_export_buffer(gpu, d_poses, handle);
		s = aclhip_peer_open_buffer(gpu, handle, &peer);
		s = aclhip_push_poses_to_peer(gpu, peer, 0, d_poses, (uint64_t)num_instances * max_tracks * 48, hip_stream);
		s = aclhip_peer_close_buffer(gpu, peer);
	}

	(void)aclhip_unregister_clip(gpu, db_clip);
	(void)aclhip_unregister_database(gpu, db);
	(void)aclhip_unregister_clip(gpu, clip);
	aclhip_destroy(gpu);
	return s == ACLHIP_OK ? 0 : 2;
}
