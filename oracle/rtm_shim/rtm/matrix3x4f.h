#pragma once
// TEST INFRASTRUCTURE (oracle/): Realtime Math stand-in, SSE2 flavour. See impl/detect_compiler.h.
// Affine 3x4 matrices with row vectors (point * matrix), as used by the reference's matrix based error metrics.
#include "rtm/qvvf.h"

namespace rtm
{
	inline matrix3x4f matrix_set(vector4f x_axis, vector4f y_axis, vector4f z_axis, vector4f w_axis) noexcept { return matrix3x4f{ x_axis, y_axis, z_axis, w_axis }; }
	inline matrix3x4f matrix_identity() noexcept { return matrix3x4f{ vector_set(1.0F, 0.0F, 0.0F, 0.0F), vector_set(0.0F, 1.0F, 0.0F, 0.0F), vector_set(0.0F, 0.0F, 1.0F, 0.0F), vector_set(0.0F, 0.0F, 0.0F, 1.0F) }; }

	inline matrix3x4f matrix_from_qvv(quatf rotation, vector4f translation, vector4f scale) noexcept
	{
		const float x = vector_get_x(rotation), y = vector_get_y(rotation), z = vector_get_z(rotation), w = vector_get_w(rotation);
		const float x2 = x + x, y2 = y + y, z2 = z + z;
		const float xx = x * x2, xy = x * y2, xz = x * z2, yy = y * y2, yz = y * z2, zz = z * z2, wx = w * x2, wy = w * y2, wz = w * z2;
		const vector4f x_axis = vector_mul(vector_set(1.0F - (yy + zz), xy + wz, xz - wy, 0.0F), vector_get_x(scale));
		const vector4f y_axis = vector_mul(vector_set(xy - wz, 1.0F - (xx + zz), yz + wx, 0.0F), vector_get_y(scale));
		const vector4f z_axis = vector_mul(vector_set(xz + wy, yz - wx, 1.0F - (xx + yy), 0.0F), vector_get_z(scale));
		return matrix3x4f{ x_axis, y_axis, z_axis, vector_set_w(translation, 1.0F) };
	}
	inline matrix3x4f matrix_from_qvv(const qvvf& transform) noexcept { return matrix_from_qvv(transform.rotation, transform.translation, transform.scale); }

	// point * matrix, translation included
	inline vector4f matrix_mul_point3(vector4f point, const matrix3x4f& mtx) noexcept
	{
		vector4f result = vector_mul(mtx.x_axis, vector_get_x(point));
		result = vector_mul_add(mtx.y_axis, vector_get_y(point), result);
		result = vector_mul_add(mtx.z_axis, vector_get_z(point), result);
		return vector_add(result, mtx.w_axis);
	}
	inline vector4f matrix_mul_vector3(vector4f vector, const matrix3x4f& mtx) noexcept
	{
		vector4f result = vector_mul(mtx.x_axis, vector_get_x(vector));
		result = vector_mul_add(mtx.y_axis, vector_get_y(vector), result);
		return vector_mul_add(mtx.z_axis, vector_get_z(vector), result);
	}

	// lhs first, then rhs
	inline matrix3x4f matrix_mul(const matrix3x4f& lhs, const matrix3x4f& rhs) noexcept
	{
		const vector4f x_axis = matrix_mul_vector3(lhs.x_axis, rhs);
		const vector4f y_axis = matrix_mul_vector3(lhs.y_axis, rhs);
		const vector4f z_axis = matrix_mul_vector3(lhs.z_axis, rhs);
		const vector4f w_axis = matrix_mul_point3(lhs.w_axis, rhs);
		return matrix3x4f{ x_axis, y_axis, z_axis, w_axis };
	}
}
