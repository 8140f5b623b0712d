#pragma once
// TEST INFRASTRUCTURE (oracle/): Realtime Math stand-in, SSE2 flavour. See impl/detect_compiler.h.
//
// Semantics restated from RTM v2.3's published x86 behaviour: every arithmetic helper is a single
// IEEE-754 fp32 operation per lane or an explicit, UNFUSED sequence of them (RTM does not use FMA
// on x86: vector_mul_add is mul then add). Composite helpers (lerp, reciprocal) document their form.
#include "rtm/types.h"
#include "rtm/scalarf.h"
#include <cstring>

#define RTM_MATRIXF_TRANSPOSE_4X4(input_xyzw0, input_xyzw1, input_xyzw2, input_xyzw3, output_xxxx, output_yyyy, output_zzzz, output_wwww) \
	do { \
		const __m128 rtm_t_x0y0x1y1 = _mm_shuffle_ps((input_xyzw0), (input_xyzw1), _MM_SHUFFLE(1, 0, 1, 0)); \
		const __m128 rtm_t_z0w0z1w1 = _mm_shuffle_ps((input_xyzw0), (input_xyzw1), _MM_SHUFFLE(3, 2, 3, 2)); \
		const __m128 rtm_t_x2y2x3y3 = _mm_shuffle_ps((input_xyzw2), (input_xyzw3), _MM_SHUFFLE(1, 0, 1, 0)); \
		const __m128 rtm_t_z2w2z3w3 = _mm_shuffle_ps((input_xyzw2), (input_xyzw3), _MM_SHUFFLE(3, 2, 3, 2)); \
		(output_xxxx) = _mm_shuffle_ps(rtm_t_x0y0x1y1, rtm_t_x2y2x3y3, _MM_SHUFFLE(2, 0, 2, 0)); \
		(output_yyyy) = _mm_shuffle_ps(rtm_t_x0y0x1y1, rtm_t_x2y2x3y3, _MM_SHUFFLE(3, 1, 3, 1)); \
		(output_zzzz) = _mm_shuffle_ps(rtm_t_z0w0z1w1, rtm_t_z2w2z3w3, _MM_SHUFFLE(2, 0, 2, 0)); \
		(output_wwww) = _mm_shuffle_ps(rtm_t_z0w0z1w1, rtm_t_z2w2z3w3, _MM_SHUFFLE(3, 1, 3, 1)); \
	} while (0)

namespace rtm
{
	//////////////////////////////////////////////////////////////////////////
	// Setters, getters, loads and stores

	inline vector4f vector_set(float x, float y, float z, float w) noexcept { return _mm_set_ps(w, z, y, x); }
	inline vector4f vector_set(float x, float y, float z) noexcept { return _mm_set_ps(0.0F, z, y, x); }
	inline vector4f vector_set(float xyzw) noexcept { return _mm_set_ps1(xyzw); }
	inline vector4f vector_set(scalarf xyzw) noexcept { return _mm_shuffle_ps(xyzw.value, xyzw.value, _MM_SHUFFLE(0, 0, 0, 0)); }
	inline vector4f vector_zero() noexcept { return _mm_setzero_ps(); }

	inline vector4f vector_load(const float* input) noexcept { return _mm_loadu_ps(input); }
	inline vector4f vector_load(const uint8_t* input) noexcept { vector4f r; std::memcpy(&r, input, sizeof(vector4f)); return r; }
	inline vector4f vector_load(const float4f* input) noexcept { return _mm_loadu_ps(&input->x); }
	inline vector4f vector_load1(const float* input) noexcept { return _mm_load_ps1(input); }
	inline vector4f vector_load2(const float* input) noexcept { return _mm_set_ps(0.0F, 0.0F, input[1], input[0]); }
	inline vector4f vector_load2(const float2f* input) noexcept { return _mm_set_ps(0.0F, 0.0F, input->y, input->x); }
	inline vector4f vector_load3(const float* input) noexcept { return _mm_set_ps(0.0F, input[2], input[1], input[0]); }
	inline vector4f vector_load3(const uint8_t* input) noexcept { float v[3]; std::memcpy(&v[0], input, sizeof(v)); return _mm_set_ps(0.0F, v[2], v[1], v[0]); }
	inline vector4f vector_load3(const float3f* input) noexcept { return _mm_set_ps(0.0F, input->z, input->y, input->x); }

	inline float vector_get_x(vector4f v) noexcept { return _mm_cvtss_f32(v); }
	inline float vector_get_y(vector4f v) noexcept { return _mm_cvtss_f32(_mm_shuffle_ps(v, v, _MM_SHUFFLE(1, 1, 1, 1))); }
	inline float vector_get_z(vector4f v) noexcept { return _mm_cvtss_f32(_mm_shuffle_ps(v, v, _MM_SHUFFLE(2, 2, 2, 2))); }
	inline float vector_get_w(vector4f v) noexcept { return _mm_cvtss_f32(_mm_shuffle_ps(v, v, _MM_SHUFFLE(3, 3, 3, 3))); }

	inline scalarf vector_get_x_as_scalar(vector4f v) noexcept { return scalarf(v); }
	inline scalarf vector_get_y_as_scalar(vector4f v) noexcept { return scalarf(_mm_shuffle_ps(v, v, _MM_SHUFFLE(1, 1, 1, 1))); }
	inline scalarf vector_get_z_as_scalar(vector4f v) noexcept { return scalarf(_mm_shuffle_ps(v, v, _MM_SHUFFLE(2, 2, 2, 2))); }
	inline scalarf vector_get_w_as_scalar(vector4f v) noexcept { return scalarf(_mm_shuffle_ps(v, v, _MM_SHUFFLE(3, 3, 3, 3))); }

	inline vector4f vector_set_x(vector4f v, float x) noexcept { return _mm_move_ss(v, _mm_set_ss(x)); }
	inline vector4f vector_set_w(vector4f v, float w) noexcept { return _mm_set_ps(w, vector_get_z(v), vector_get_y(v), vector_get_x(v)); }

	inline void vector_store(vector4f v, float* output) noexcept { _mm_storeu_ps(output, v); }
	inline void vector_store(vector4f v, uint8_t* output) noexcept { std::memcpy(output, &v, sizeof(vector4f)); }
	inline void vector_store(vector4f v, float4f* output) noexcept { _mm_storeu_ps(&output->x, v); }
	inline void vector_store2(vector4f v, float* output) noexcept { output[0] = vector_get_x(v); output[1] = vector_get_y(v); }
	inline void vector_store2(vector4f v, uint8_t* output) noexcept { std::memcpy(output, &v, sizeof(float) * 2); }
	inline void vector_store2(vector4f v, float2f* output) noexcept { output->x = vector_get_x(v); output->y = vector_get_y(v); }
	inline void vector_store3(vector4f v, float* output) noexcept { output[0] = vector_get_x(v); output[1] = vector_get_y(v); output[2] = vector_get_z(v); }
	inline void vector_store3(vector4f v, uint8_t* output) noexcept { std::memcpy(output, &v, sizeof(float) * 3); }
	inline void vector_store3(vector4f v, float3f* output) noexcept { output->x = vector_get_x(v); output->y = vector_get_y(v); output->z = vector_get_z(v); }

	inline vector4f quat_to_vector(quatf q) noexcept { return q; }
	inline quatf vector_to_quat(vector4f v) noexcept { return v; }

	//////////////////////////////////////////////////////////////////////////
	// Arithmetic (one IEEE op per lane unless stated)

	inline vector4f vector_add(vector4f a, vector4f b) noexcept { return _mm_add_ps(a, b); }
	inline vector4f vector_sub(vector4f a, vector4f b) noexcept { return _mm_sub_ps(a, b); }
	inline vector4f vector_mul(vector4f a, vector4f b) noexcept { return _mm_mul_ps(a, b); }
	inline vector4f vector_mul(vector4f a, float b) noexcept { return _mm_mul_ps(a, _mm_set_ps1(b)); }
	inline vector4f vector_div(vector4f a, vector4f b) noexcept { return _mm_div_ps(a, b); }
	inline vector4f vector_min(vector4f a, vector4f b) noexcept { return _mm_min_ps(a, b); }
	inline vector4f vector_max(vector4f a, vector4f b) noexcept { return _mm_max_ps(a, b); }
	inline vector4f vector_clamp(vector4f v, vector4f lo, vector4f hi) noexcept { return _mm_min_ps(hi, _mm_max_ps(lo, v)); }
	inline vector4f vector_sqrt(vector4f v) noexcept { return _mm_sqrt_ps(v); }
	inline vector4f vector_neg(vector4f v) noexcept { return _mm_xor_ps(v, _mm_set_ps1(-0.0F)); }
	inline vector4f vector_abs(vector4f v) noexcept { return _mm_and_ps(v, _mm_castsi128_ps(_mm_set1_epi32(0x7FFFFFFF))); }
	inline vector4f vector_and(vector4f a, vector4f b) noexcept { return _mm_and_ps(a, b); }
	inline vector4f vector_or(vector4f a, vector4f b) noexcept { return _mm_or_ps(a, b); }
	inline vector4f vector_xor(vector4f a, vector4f b) noexcept { return _mm_xor_ps(a, b); }

	// v0 * v1 + v2 as an UNFUSED multiply then add (RTM's x86 path).
	inline vector4f vector_mul_add(vector4f v0, vector4f v1, vector4f v2) noexcept { return _mm_add_ps(_mm_mul_ps(v0, v1), v2); }
	inline vector4f vector_mul_add(vector4f v0, float s1, vector4f v2) noexcept { return _mm_add_ps(_mm_mul_ps(v0, _mm_set_ps1(s1)), v2); }
	// v2 - v0 * v1 as an UNFUSED multiply then subtract.
	inline vector4f vector_neg_mul_sub(vector4f v0, vector4f v1, vector4f v2) noexcept { return _mm_sub_ps(v2, _mm_mul_ps(v0, v1)); }
	inline vector4f vector_neg_mul_sub(vector4f v0, float s1, vector4f v2) noexcept { return _mm_sub_ps(v2, _mm_mul_ps(v0, _mm_set_ps1(s1))); }

	// Stable lerp: (start - alpha * start) + alpha * end; exact at alpha 0 and 1 (what the reference's
	// own validator requires: tools/acl_compressor/sources/validate_tracks.cpp:117,189-211).
	inline vector4f vector_lerp(vector4f start, vector4f end, float alpha) noexcept { return vector_mul_add(end, alpha, vector_neg_mul_sub(start, alpha, start)); }
	inline vector4f vector_lerp(vector4f start, vector4f end, vector4f alpha) noexcept { return vector_mul_add(end, alpha, vector_neg_mul_sub(start, alpha, start)); }

	inline vector4f vector_lerp(vector4f start, vector4f end, scalarf alpha) noexcept { return vector_lerp(start, end, vector_set(alpha)); }

	// Exact division form (RTM refines a hardware estimate; only used off the decode hot path).
	inline vector4f vector_reciprocal(vector4f v) noexcept { return _mm_div_ps(_mm_set_ps1(1.0F), v); }

	inline vector4f vector_floor(vector4f v) noexcept { return _mm_set_ps(std::floor(vector_get_w(v)), std::floor(vector_get_z(v)), std::floor(vector_get_y(v)), std::floor(vector_get_x(v))); }
	inline vector4f vector_ceil(vector4f v) noexcept { return _mm_set_ps(std::ceil(vector_get_w(v)), std::ceil(vector_get_z(v)), std::ceil(vector_get_y(v)), std::ceil(vector_get_x(v))); }
	inline vector4f vector_round_symmetric(vector4f v) noexcept { return _mm_set_ps(scalar_round_symmetric(vector_get_w(v)), scalar_round_symmetric(vector_get_z(v)), scalar_round_symmetric(vector_get_y(v)), scalar_round_symmetric(vector_get_x(v))); }

	inline float vector_dot(vector4f a, vector4f b) noexcept
	{
		const __m128 x2_y2_z2_w2 = _mm_mul_ps(a, b);
		const __m128 z2_w2_0_0 = _mm_shuffle_ps(x2_y2_z2_w2, x2_y2_z2_w2, _MM_SHUFFLE(0, 0, 3, 2));
		const __m128 x2z2_y2w2_0_0 = _mm_add_ps(x2_y2_z2_w2, z2_w2_0_0);
		const __m128 y2w2_0_0_0 = _mm_shuffle_ps(x2z2_y2w2_0_0, x2z2_y2w2_0_0, _MM_SHUFFLE(0, 0, 0, 1));
		return _mm_cvtss_f32(_mm_add_ps(x2z2_y2w2_0_0, y2w2_0_0_0));
	}
	inline float vector_dot3(vector4f a, vector4f b) noexcept
	{
		const __m128 m = _mm_mul_ps(a, b);
		return (vector_get_x(m) + vector_get_y(m)) + vector_get_z(m);
	}
	inline float vector_length_squared3(vector4f v) noexcept { return vector_dot3(v, v); }
	inline float vector_length3(vector4f v) noexcept { return std::sqrt(vector_length_squared3(v)); }
	inline float vector_length3_as_scalar(vector4f v) noexcept { return vector_length3(v); }
	inline float vector_distance3(vector4f a, vector4f b) noexcept { return vector_length3(_mm_sub_ps(b, a)); }
	inline float vector_distance3_as_scalar(vector4f a, vector4f b) noexcept { return vector_distance3(a, b); }
	inline vector4f vector_cross3(vector4f a, vector4f b) noexcept
	{
		const float ax = vector_get_x(a), ay = vector_get_y(a), az = vector_get_z(a);
		const float bx = vector_get_x(b), by = vector_get_y(b), bz = vector_get_z(b);
		return vector_set((ay * bz) - (az * by), (az * bx) - (ax * bz), (ax * by) - (ay * bx), 0.0F);
	}
	inline float vector_get_max_component(vector4f v) noexcept { return scalar_max(scalar_max(vector_get_x(v), vector_get_y(v)), scalar_max(vector_get_z(v), vector_get_w(v))); }
	inline float vector_get_min_component(vector4f v) noexcept { return scalar_min(scalar_min(vector_get_x(v), vector_get_y(v)), scalar_min(vector_get_z(v), vector_get_w(v))); }

	//////////////////////////////////////////////////////////////////////////
	// Comparisons and masks

	inline mask4f mask_set(bool x, bool y, bool z, bool w) noexcept { return _mm_castsi128_ps(_mm_set_epi32(-int32_t(w), -int32_t(z), -int32_t(y), -int32_t(x))); }
	inline mask4f vector_less_than(vector4f a, vector4f b) noexcept { return _mm_cmplt_ps(a, b); }
	inline mask4f vector_less_equal(vector4f a, vector4f b) noexcept { return _mm_cmple_ps(a, b); }
	inline mask4f vector_greater_than(vector4f a, vector4f b) noexcept { return _mm_cmpgt_ps(a, b); }
	inline mask4f vector_greater_equal(vector4f a, vector4f b) noexcept { return _mm_cmpge_ps(a, b); }
	inline mask4f vector_equal(vector4f a, vector4f b) noexcept { return _mm_cmpeq_ps(a, b); }

	// Per lane: mask ? if_true : if_false
	inline vector4f vector_select(mask4f mask, vector4f if_true, vector4f if_false) noexcept { return _mm_or_ps(_mm_andnot_ps(mask, if_false), _mm_and_ps(if_true, mask)); }

	inline bool vector_all_less_than(vector4f a, vector4f b) noexcept { return _mm_movemask_ps(_mm_cmplt_ps(a, b)) == 0xF; }
	inline bool vector_all_less_equal(vector4f a, vector4f b) noexcept { return _mm_movemask_ps(_mm_cmple_ps(a, b)) == 0xF; }
	inline bool vector_all_less_equal3(vector4f a, vector4f b) noexcept { return (_mm_movemask_ps(_mm_cmple_ps(a, b)) & 0x7) == 0x7; }
	inline bool vector_all_greater_equal(vector4f a, vector4f b) noexcept { return _mm_movemask_ps(_mm_cmpge_ps(a, b)) == 0xF; }
	inline bool vector_all_greater_equal3(vector4f a, vector4f b) noexcept { return (_mm_movemask_ps(_mm_cmpge_ps(a, b)) & 0x7) == 0x7; }
	inline bool vector_all_equal(vector4f a, vector4f b) noexcept { return _mm_movemask_ps(_mm_cmpeq_ps(a, b)) == 0xF; }
	inline bool vector_all_equal3(vector4f a, vector4f b) noexcept { return (_mm_movemask_ps(_mm_cmpeq_ps(a, b)) & 0x7) == 0x7; }
	inline bool vector_all_near_equal(vector4f a, vector4f b, float threshold = 0.00001F) noexcept { return vector_all_less_equal(vector_abs(_mm_sub_ps(a, b)), _mm_set_ps1(threshold)); }
	inline bool vector_all_near_equal3(vector4f a, vector4f b, float threshold = 0.00001F) noexcept { return vector_all_less_equal3(vector_abs(_mm_sub_ps(a, b)), _mm_set_ps1(threshold)); }
	inline bool vector_is_finite(vector4f v) noexcept { return std::isfinite(vector_get_x(v)) && std::isfinite(vector_get_y(v)) && std::isfinite(vector_get_z(v)) && std::isfinite(vector_get_w(v)); }
	inline bool vector_is_finite2(vector4f v) noexcept { return std::isfinite(vector_get_x(v)) && std::isfinite(vector_get_y(v)); }
	inline bool vector_is_finite3(vector4f v) noexcept { return std::isfinite(vector_get_x(v)) && std::isfinite(vector_get_y(v)) && std::isfinite(vector_get_z(v)); }

	//////////////////////////////////////////////////////////////////////////
	// Swizzle: result.{x,y} come from (input0|input1) per comp0/comp1, likewise {z,w}; mix4::a..d select input1.xyzw
	namespace rtm_impl
	{
		inline float mix_component(vector4f input0, vector4f input1, mix4 c) noexcept
		{
			alignas(16) float v0[4]; alignas(16) float v1[4];
			_mm_store_ps(v0, input0); _mm_store_ps(v1, input1);
			const int i = static_cast<int>(c);
			return i < 4 ? v0[i] : v1[i - 4];
		}
	}
	template<mix4 comp0, mix4 comp1, mix4 comp2, mix4 comp3>
	inline vector4f vector_mix(vector4f input0, vector4f input1) noexcept
	{
		return vector_set(rtm_impl::mix_component(input0, input1, comp0), rtm_impl::mix_component(input0, input1, comp1),
			rtm_impl::mix_component(input0, input1, comp2), rtm_impl::mix_component(input0, input1, comp3));
	}
	inline vector4f vector_dup_x(vector4f v) noexcept { return _mm_shuffle_ps(v, v, _MM_SHUFFLE(0, 0, 0, 0)); }
	inline vector4f vector_dup_y(vector4f v) noexcept { return _mm_shuffle_ps(v, v, _MM_SHUFFLE(1, 1, 1, 1)); }
	inline vector4f vector_dup_z(vector4f v) noexcept { return _mm_shuffle_ps(v, v, _MM_SHUFFLE(2, 2, 2, 2)); }
	inline vector4f vector_dup_w(vector4f v) noexcept { return _mm_shuffle_ps(v, v, _MM_SHUFFLE(3, 3, 3, 3)); }
}
