#pragma once
// TEST INFRASTRUCTURE (oracle/): Realtime Math stand-in, SSE2 flavour. See impl/detect_compiler.h.
#include "rtm/quatf.h"

namespace rtm
{
	// w = sqrt(|((1 - x*x) - y*y) - z*z|): this operation order is RTM's documented one.
	inline quatf quat_from_positive_w(vector4f input) noexcept
	{
		const __m128 x2y2z2 = _mm_mul_ps(input, input);
		const __m128 one = _mm_set_ss(1.0F);
		__m128 w_squared = _mm_sub_ss(_mm_sub_ss(_mm_sub_ss(one, x2y2z2), _mm_shuffle_ps(x2y2z2, x2y2z2, _MM_SHUFFLE(1, 1, 1, 1))), _mm_shuffle_ps(x2y2z2, x2y2z2, _MM_SHUFFLE(2, 2, 2, 2)));
		w_squared = _mm_andnot_ps(_mm_set_ss(-0.0F), w_squared);
		const __m128 w = _mm_sqrt_ss(w_squared);
		return _mm_set_ps(_mm_cvtss_f32(w), vector_get_z(input), vector_get_y(input), vector_get_x(input));
	}

	// Negates the whole quaternion when W is negative.
	inline quatf quat_ensure_positive_w(quatf input) noexcept
	{
		const __m128 w = _mm_shuffle_ps(input, input, _MM_SHUFFLE(3, 3, 3, 3));
		const __m128 sign = _mm_and_ps(w, _mm_set_ps1(-0.0F));
		return _mm_xor_ps(input, sign);
	}
}
