#pragma once
// TEST INFRASTRUCTURE (oracle/): Realtime Math stand-in, SSE2 flavour. See impl/detect_compiler.h.
#include "rtm/quatf.h"

namespace rtm
{
	inline qvvf qvv_set(quatf rotation, vector4f translation, vector4f scale) noexcept { return qvvf{ rotation, translation, scale }; }
	inline qvvf qvv_identity() noexcept { return qvvf{ quat_identity(), vector_zero(), vector_set(1.0F) }; }
	inline bool qvv_is_finite(const qvvf& t) noexcept { return quat_is_finite(t.rotation) && vector_is_finite3(t.translation) && vector_is_finite3(t.scale); }

	namespace rtm_impl
	{
		// 1 / sqrt(x) the way RTM's x86 scalar_sqrt_reciprocal does it: RSQRTSS refined by two Newton-Raphson steps
		inline float sqrt_reciprocal(float input) noexcept { return _mm_cvtss_f32(rsqrt_nr2_ss(_mm_set_ss(input))); }

		// RTM's route through 3x4 matrices for negative scales (rtm/qvvf.h qvv_mul): matrix_from_qvv x 2, matrix_mul,
		// matrix_remove_scale, axes * sign(scale), quat_from_matrix. Row vectors; lhs first.
		inline void rotation_scale_rows(const qvvf& t, float m[3][3]) noexcept
		{
			const float x = vector_get_x(t.rotation), y = vector_get_y(t.rotation), z = vector_get_z(t.rotation), w = vector_get_w(t.rotation);
			const float sx = vector_get_x(t.scale), sy = vector_get_y(t.scale), sz = vector_get_z(t.scale);
			const float x2 = x + x, y2 = y + y, z2 = z + z;
			const float xx = x * x2, xy = x * y2, xz = x * z2, yy = y * y2, yz = y * z2, zz = z * z2, wx = w * x2, wy = w * y2, wz = w * z2;
			m[0][0] = (1.0F - (yy + zz)) * sx; m[0][1] = (xy + wz) * sx; m[0][2] = (xz - wy) * sx;
			m[1][0] = (xy - wz) * sy; m[1][1] = (1.0F - (xx + zz)) * sy; m[1][2] = (yz + wx) * sy;
			m[2][0] = (xz + wy) * sz; m[2][1] = (yz - wx) * sz; m[2][2] = (1.0F - (xx + yy)) * sz;
		}

		inline quatf quat_from_rows(const float m[3][3]) noexcept
		{
			float q[4];
			const float trace = (m[0][0] + m[1][1]) + m[2][2];
			if (trace > 0.0F)
			{
				const float inv_trace = sqrt_reciprocal(trace + 1.0F);
				const float half_inv_trace = inv_trace * 0.5F;
				q[0] = (m[1][2] - m[2][1]) * half_inv_trace;
				q[1] = (m[2][0] - m[0][2]) * half_inv_trace;
				q[2] = (m[0][1] - m[1][0]) * half_inv_trace;
				q[3] = (1.0F / inv_trace) * 0.5F;
			}
			else
			{
				int best = 0;
				if (m[1][1] > m[0][0])
					best = 1;
				if (m[2][2] > m[best][best])
					best = 2;
				const int next = (best + 1) % 3, last = (next + 1) % 3;
				const float pseudo_trace = ((1.0F + m[best][best]) - m[next][next]) - m[last][last];
				const float inv_pseudo_trace = sqrt_reciprocal(pseudo_trace);
				const float half_inv_pseudo_trace = inv_pseudo_trace * 0.5F;
				q[best] = (1.0F / inv_pseudo_trace) * 0.5F;
				q[next] = half_inv_pseudo_trace * (m[best][next] + m[next][best]);
				q[last] = half_inv_pseudo_trace * (m[best][last] + m[last][best]);
				q[3] = half_inv_pseudo_trace * (m[next][last] - m[last][next]);
			}
			return quat_normalize(quat_set(q[0], q[1], q[2], q[3]));
		}

		inline qvvf qvv_mul_through_matrices(const qvvf& lhs, const qvvf& rhs) noexcept
		{
			float l[3][3], r[3][3], axes[3][3], translation[3], scale[3];
			rotation_scale_rows(lhs, l);
			rotation_scale_rows(rhs, r);
			const float lt[3] = { vector_get_x(lhs.translation), vector_get_y(lhs.translation), vector_get_z(lhs.translation) };
			const float rt[3] = { vector_get_x(rhs.translation), vector_get_y(rhs.translation), vector_get_z(rhs.translation) };
			const float ls[3] = { vector_get_x(lhs.scale), vector_get_y(lhs.scale), vector_get_z(lhs.scale) };
			const float rs[3] = { vector_get_x(rhs.scale), vector_get_y(rhs.scale), vector_get_z(rhs.scale) };
			for (int c = 0; c < 3; ++c)
				translation[c] = rt[c] + (((lt[0] * r[0][c]) + (lt[1] * r[1][c])) + (lt[2] * r[2][c]));
			for (int row = 0; row < 3; ++row)
			{
				float axis[3];
				for (int c = 0; c < 3; ++c)
					axis[c] = ((l[row][0] * r[0][c]) + (l[row][1] * r[1][c])) + (l[row][2] * r[2][c]);
				const float length_squared = ((axis[0] * axis[0]) + (axis[1] * axis[1])) + (axis[2] * axis[2]);
				scale[row] = ls[row] * rs[row];
				const float sign = scale[row] >= 0.0F ? 1.0F : -1.0F;
				const float inv_length = length_squared >= 1.0E-8F ? sqrt_reciprocal(length_squared) : 1.0F;
				for (int c = 0; c < 3; ++c)
					axes[row][c] = (length_squared >= 1.0E-8F ? axis[c] * inv_length : axis[c]) * sign;
			}
			return qvvf{ quat_from_rows(axes), vector_set(translation[0], translation[1], translation[2], 0.0F), vector_set(scale[0], scale[1], scale[2], 0.0F) };
		}
	}

	// Applies lhs first, then rhs (child-then-parent), scale aware. Negative scales cannot ride on a quaternion: RTM composes
	// matrices for them.
	inline qvvf qvv_mul(const qvvf& lhs, const qvvf& rhs) noexcept
	{
		const vector4f min_scale = vector_min(lhs.scale, rhs.scale);
		if (vector_get_x(min_scale) < 0.0F || vector_get_y(min_scale) < 0.0F || vector_get_z(min_scale) < 0.0F)
			return rtm_impl::qvv_mul_through_matrices(lhs, rhs);
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
