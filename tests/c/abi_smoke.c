/* The C ABI from a plain C99 translation unit (gcc -std=c99 -pedantic): the header must not need C++, and the host only entry
 * points must work without a GPU. argv[1]: a file holding a compressed_tracks blob, argv[2]: "valid" or "invalid". */
#include "aclhip.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

int main(int argc, char** argv)
{
	aclhip_decompress_params params;
	char message[256];
	unsigned char* blob;
	long size;
	FILE* file;
	aclhip_status status;

	if (argc != 3)
		return 2;

	if (aclhip_abi_version() != ACLHIP_ABI_VERSION)
		return 7;		/* the library was built from another header than this translation unit */
	aclhip_default_params(&params);
	if (params.rounding_policy != ACLHIP_ROUND_NONE || params.looping_policy != ACLHIP_LOOP_AS_COMPRESSED || params.normalization != ACLHIP_NORMALIZE_LERP_ONLY)
		return 3;		/* default_transform_decompression_settings + as_compressed looping */
	if (strcmp(aclhip_status_string(ACLHIP_OK), aclhip_status_string(ACLHIP_ERROR_INVALID_CLIP)) == 0)
		return 4;

	/* the other host only entry points: a decode order, a walk schedule (a chain of four under one root: three steps) */
	{
		const aclhip_clip clips[6] = { 9, 1, 9, 8, 1, 9 };
		const uint32_t parents[5] = { ACLHIP_NO_PARENT, 0, 1, 2, 0 };
		uint32_t order[6], steps[5], num_steps = 0, seen = 0, i;
		if (aclhip_order_instances_for_locality(NULL, clips, 6, order) != ACLHIP_OK)
			return 20;
		for (i = 0; i < 6; ++i)
			seen |= 1u << order[i];
		if (seen != 63u)
			return 21;		/* a permutation */
		if (aclhip_plan_hierarchy_walk(parents, 5, 8, steps, &num_steps) != ACLHIP_OK || num_steps != 3 || steps[0] != 0 || steps[1] != 1 || steps[4] != 1 || steps[3] != 3)
			return 22;
	}

	file = fopen(argv[1], "rb");
	if (file == NULL)
		return 5;
	fseek(file, 0, SEEK_END);
	size = ftell(file);
	fseek(file, 0, SEEK_SET);
	blob = (unsigned char*)malloc((size_t)size + 16);
	if (blob == NULL || fread(blob, 1, (size_t)size, file) != (size_t)size)
		return 6;
	fclose(file);

	status = aclhip_check_clip(blob, (unsigned long long)size, 1, message, sizeof(message));
	printf("%d %s\n", (int)status, message);
	free(blob);
	if (strcmp(argv[2], "valid") == 0)
		return status == ACLHIP_OK ? 0 : 10;
	return status != ACLHIP_OK && message[0] != '\0' ? 0 : 11;
}
