// TEST INFRASTRUCTURE -- not part of the product.
//
// The reference's own pose consumers, unmodified headers read in place from /root/reference and compiled against oracle/rtm_shim/:
// acl::apply_additive_to_base (core/additive_utils.h:150) and acl::local_to_object_space (compression/transform_pose_utils.h:35)
// over poses of rtm::qvvf (48 bytes: rotation | translation | scale). Output: oracle/_ref/libaclref_pose.so.
#include <acl/core/additive_utils.h>
#include <acl/compression/transform_pose_utils.h>

#include <cstdint>
#include <cstring>
#include <vector>

static_assert(sizeof(rtm::qvvf) == 48, "pose layout");

extern "C" void aclref_apply_additive_to_base(int additive_format, const float* base_pose, const float* additive_pose, uint32_t num_transforms, float* out_pose)
{
	for (uint32_t i = 0; i < num_transforms; ++i)
	{
		rtm::qvvf base, additive;
		std::memcpy(&base, base_pose + size_t(i) * 12, sizeof(base));
		std::memcpy(&additive, additive_pose + size_t(i) * 12, sizeof(additive));
		const rtm::qvvf result = acl::apply_additive_to_base(static_cast<acl::additive_clip_format8>(additive_format), base, additive);
		std::memcpy(out_pose + size_t(i) * 12, &result, sizeof(result));
	}
}

extern "C" void aclref_local_to_object_space(const uint32_t* parent_indices, const float* local_pose, uint32_t num_transforms, float* out_object_pose)
{
	std::vector<rtm::qvvf> local(num_transforms), object(num_transforms);
	if (num_transforms != 0)
		std::memcpy(local.data(), local_pose, size_t(num_transforms) * sizeof(rtm::qvvf));
	acl::local_to_object_space(parent_indices, local.data(), num_transforms, object.data());
	if (num_transforms != 0)
		std::memcpy(out_object_pose, object.data(), size_t(num_transforms) * sizeof(rtm::qvvf));
}
