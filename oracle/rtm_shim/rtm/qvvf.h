#pragma once
// TEST INFRASTRUCTURE (oracle/): Realtime Math stand-in, SSE2 flavour. See impl/detect_compiler.h.
#include "rtm/quatf.h"

namespace rtm
{
	inline qvvf qvv_set(quatf rotation, vector4f translation, vector4f scale) noexcept { return qvvf{ rotation, translation, scale }; }
	inline qvvf qvv_identity() noexcept { return qvvf{ quat_identity(), vector_zero(), vector_set(1.0F) }; }
	inline bool qvv_is_finite(const qvvf& t) noexcept { return quat_is_finite(t.rotation) && vector_is_finite3(t.translation) && vector_is_finite3(t.scale); }

	// Applies lhs first, then rhs (child-then-parent), scale aware.
	inline qvvf qvv_mul(const qvvf& lhs, const qvvf& rhs) noexcept
	{
		const quatf rotation = quat_mul(lhs.rotation, rhs.rotation);
		const vector4f translation = vector_add(quat_mul_vector3(vector_mul(lhs.translation, rhs.scale), rhs.rotation), rhs.translation);
		const vector4f scale = vector_mul(lhs.scale, rhs.scale);
		return qvv_set(rotation, translation, scale);
	}
	inline qvvf qvv_mul_no_scale(const qvvf& lhs, const qvvf& rhs) noexcept
	{
		const quatf rotation = quat_mul(lhs.rotation, rhs.rotation);
		const vector4f translation = vector_add(quat_mul_vector3(lhs.translation, rhs.rotation), rhs.translation);
		return qvv_set(rotation, translation, vector_set(1.0F));
	}
	inline vector4f qvv_mul_point3(vector4f point, const qvvf& t) noexcept { return vector_add(quat_mul_vector3(vector_mul(t.scale, point), t.rotation), t.translation); }
	inline vector4f qvv_mul_point3_no_scale(vector4f point, const qvvf& t) noexcept { return vector_add(quat_mul_vector3(point, t.rotation), t.translation); }
	inline qvvf qvv_inverse(const qvvf& input) noexcept
	{
		const quatf inv_rotation = quat_conjugate(input.rotation);
		const vector4f inv_scale = vector_reciprocal(input.scale);
		const vector4f inv_translation = vector_neg(quat_mul_vector3(vector_mul(inv_scale, input.translation), inv_rotation));
		return qvv_set(inv_rotation, inv_translation, inv_scale);
	}
	inline qvvf qvv_normalize(const qvvf& input) noexcept { return qvv_set(quat_normalize(input.rotation), input.translation, input.scale); }
}
