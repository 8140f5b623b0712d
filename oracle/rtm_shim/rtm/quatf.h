#pragma once
// TEST INFRASTRUCTURE (oracle/): Realtime Math stand-in, SSE2 flavour. See impl/detect_compiler.h.
#include "rtm/vector4f.h"

namespace rtm
{
	inline quatf quat_identity() noexcept { return _mm_set_ps(1.0F, 0.0F, 0.0F, 0.0F); }
	inline quatf quat_set(float x, float y, float z, float w) noexcept { return _mm_set_ps(w, z, y, x); }
	inline quatf quat_load(const float* input) noexcept { return _mm_loadu_ps(input); }
	inline quatf quat_load(const float4f* input) noexcept { return _mm_loadu_ps(&input->x); }
	inline void quat_store(quatf q, float* output) noexcept { _mm_storeu_ps(output, q); }
	inline void quat_store(quatf q, uint8_t* output) noexcept { std::memcpy(output, &q, sizeof(quatf)); }
	inline void quat_store(quatf q, float4f* output) noexcept { _mm_storeu_ps(&output->x, q); }

	inline float quat_get_x(quatf q) noexcept { return vector_get_x(q); }
	inline float quat_get_y(quatf q) noexcept { return vector_get_y(q); }
	inline float quat_get_z(quatf q) noexcept { return vector_get_z(q); }
	inline float quat_get_w(quatf q) noexcept { return vector_get_w(q); }

	inline quatf quat_conjugate(quatf q) noexcept { return _mm_xor_ps(q, _mm_set_ps(0.0F, -0.0F, -0.0F, -0.0F)); }

	// Hamilton product in RTM's convention: quat_mul(lhs, rhs) applies lhs first, then rhs. The four products of every lane are
	// summed pairwise, (a*rw + b*rx) + (c*ry + d*rz), with the signs applied to the products before the additions: the
	// association of RTM's SSE2 form (whole-register multiplies by rhs.wwww / xxxx / yyyy / zzzz, sign flips, two adds, one add).
	inline quatf quat_mul(quatf lhs, quatf rhs) noexcept
	{
		const __m128 sign_wzyx = _mm_set_ps(-0.0F, 0.0F, -0.0F, 0.0F);		// lanes x, y, z, w = +, -, +, -
		const __m128 sign_zwxy = _mm_set_ps(-0.0F, -0.0F, 0.0F, 0.0F);		// +, +, -, -
		const __m128 sign_yxwz = _mm_set_ps(-0.0F, 0.0F, 0.0F, -0.0F);		// -, +, +, -

		const __m128 r_xxxx = _mm_shuffle_ps(rhs, rhs, _MM_SHUFFLE(0, 0, 0, 0));
		const __m128 r_yyyy = _mm_shuffle_ps(rhs, rhs, _MM_SHUFFLE(1, 1, 1, 1));
		const __m128 r_zzzz = _mm_shuffle_ps(rhs, rhs, _MM_SHUFFLE(2, 2, 2, 2));
		const __m128 r_wwww = _mm_shuffle_ps(rhs, rhs, _MM_SHUFFLE(3, 3, 3, 3));

		const __m128 l_wzyx = _mm_shuffle_ps(lhs, lhs, _MM_SHUFFLE(0, 1, 2, 3));
		const __m128 l_zwxy = _mm_shuffle_ps(lhs, lhs, _MM_SHUFFLE(1, 0, 3, 2));
		const __m128 l_yxwz = _mm_shuffle_ps(lhs, lhs, _MM_SHUFFLE(2, 3, 0, 1));

		const __m128 by_w = _mm_mul_ps(r_wwww, lhs);
		const __m128 by_x = _mm_xor_ps(_mm_mul_ps(r_xxxx, l_wzyx), sign_wzyx);
		const __m128 by_y = _mm_xor_ps(_mm_mul_ps(r_yyyy, l_zwxy), sign_zwxy);
		const __m128 by_z = _mm_xor_ps(_mm_mul_ps(r_zzzz, l_yxwz), sign_yxwz);
		return _mm_add_ps(_mm_add_ps(by_w, by_x), _mm_add_ps(by_y, by_z));
	}

	// Rotates a vector3: q^-1 * v * q in RTM's multiplication convention.
	inline vector4f quat_mul_vector3(vector4f vector, quatf rotation) noexcept
	{
		const quatf vector_quat = vector_set_w(vector, 0.0F);
		const quatf inv_rotation = quat_conjugate(rotation);
		return quat_mul(quat_mul(inv_rotation, vector_quat), rotation);
	}

	inline float quat_length_squared(quatf q) noexcept { return vector_dot(q, q); }
	inline float quat_length(quatf q) noexcept { return std::sqrt(quat_length_squared(q)); }

	namespace rtm_impl
	{
		// 1/sqrt(x) in lane 0: hardware estimate refined by two Newton-Raphson steps (RTM's x86 form).
		inline __m128 rsqrt_nr2_ss(__m128 input) noexcept
		{
			const __m128 half = _mm_set_ss(0.5F);
			const __m128 input_half = _mm_mul_ss(input, half);
			const __m128 x0 = _mm_rsqrt_ss(input);
			__m128 x1 = _mm_mul_ss(x0, x0);
			x1 = _mm_sub_ss(half, _mm_mul_ss(input_half, x1));
			x1 = _mm_add_ss(_mm_mul_ss(x0, x1), x0);
			__m128 x2 = _mm_mul_ss(x1, x1);
			x2 = _mm_sub_ss(half, _mm_mul_ss(input_half, x2));
			x2 = _mm_add_ss(_mm_mul_ss(x1, x2), x1);
			return x2;
		}

		inline __m128 dot4_ss(__m128 a, __m128 b) noexcept
		{
			const __m128 x2_y2_z2_w2 = _mm_mul_ps(a, b);
			const __m128 z2_w2_0_0 = _mm_shuffle_ps(x2_y2_z2_w2, x2_y2_z2_w2, _MM_SHUFFLE(0, 0, 3, 2));
			const __m128 x2z2_y2w2_0_0 = _mm_add_ps(x2_y2_z2_w2, z2_w2_0_0);
			const __m128 y2w2_0_0_0 = _mm_shuffle_ps(x2z2_y2w2_0_0, x2z2_y2w2_0_0, _MM_SHUFFLE(0, 0, 0, 1));
			return _mm_add_ps(x2z2_y2w2_0_0, y2w2_0_0_0);
		}
	}

	inline quatf quat_normalize(quatf input) noexcept
	{
		const __m128 len_sq = rtm_impl::dot4_ss(input, input);
		const __m128 inv_len_ss = rtm_impl::rsqrt_nr2_ss(len_sq);
		const __m128 inv_len = _mm_shuffle_ps(inv_len_ss, inv_len_ss, _MM_SHUFFLE(0, 0, 0, 0));
		return _mm_mul_ps(input, inv_len);
	}

	// Normalised linear interpolation along the shortest arc (sign of dot selects end or -end).
	inline quatf quat_lerp(quatf start, quatf end, float alpha) noexcept
	{
		const __m128 dot_ss = rtm_impl::dot4_ss(start, end);
		const __m128 dot = _mm_shuffle_ps(dot_ss, dot_ss, _MM_SHUFFLE(0, 0, 0, 0));
		const __m128 bias = _mm_and_ps(dot, _mm_set_ps1(-0.0F));
		const __m128 alpha_ = _mm_set_ps1(alpha);
		const __m128 interpolated = _mm_add_ps(_mm_sub_ps(start, _mm_mul_ps(alpha_, start)), _mm_mul_ps(alpha_, _mm_xor_ps(end, bias)));
		return quat_normalize(interpolated);
	}

	inline bool quat_is_finite(quatf q) noexcept { return vector_is_finite(q); }
	inline bool quat_is_normalized(quatf q, float threshold = 0.00001F) noexcept { return std::fabs(quat_length_squared(q) - 1.0F) < threshold; }
	inline bool quat_near_equal(quatf a, quatf b, float threshold = 0.00001F) noexcept { return vector_all_near_equal(a, b, threshold); }

	// True when the rotation angle is below the threshold; uses the positive-W half-angle like RTM.
	inline bool quat_near_identity(quatf q, float threshold_angle = 0.00284714461F) noexcept
	{
		const float positive_w = std::fabs(vector_get_w(q));
		const float clamped_w = positive_w < 1.0F ? positive_w : 1.0F;
		const float angle = std::acos(clamped_w) * 2.0F;
		return angle < threshold_angle;
	}
}
