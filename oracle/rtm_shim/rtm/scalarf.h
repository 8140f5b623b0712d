#pragma once
// TEST INFRASTRUCTURE (oracle/): Realtime Math stand-in. See impl/detect_compiler.h.
#include "rtm/types.h"
#include <algorithm>
#include <cmath>
#include <limits>

namespace rtm
{
	inline scalarf scalar_set(float x) noexcept { return scalarf{ _mm_set_ps1(x) }; }
	inline scalarf scalar_load_as_scalar(const float* input) noexcept { return scalarf{ _mm_load_ss(input) }; }
	inline float scalar_cast(float x) noexcept { return x; }
	inline float scalar_cast(scalarf x) noexcept { return _mm_cvtss_f32(x.value); }
	inline void scalar_store(float x, float* output) noexcept { *output = x; }
	inline void scalar_store(scalarf x, float* output) noexcept { _mm_store_ss(output, x.value); }

	inline float scalar_abs(float x) noexcept { return std::fabs(x); }
	inline float scalar_min(float a, float b) noexcept { return a < b ? a : b; }
	inline float scalar_max(float a, float b) noexcept { return a > b ? a : b; }
	inline float scalar_clamp(float x, float lo, float hi) noexcept { return scalar_min(scalar_max(x, lo), hi); }
	inline float scalar_floor(float x) noexcept { return std::floor(x); }
	inline float scalar_ceil(float x) noexcept { return std::ceil(x); }
	inline float scalar_sqrt(float x) noexcept { return std::sqrt(x); }
	inline float scalar_mul_add(float a, float b, float c) noexcept { return (a * b) + c; }
	inline float scalar_neg_mul_sub(float a, float b, float c) noexcept { return c - (a * b); }
	// Stable form: returns 'start' at alpha 0 and 'end' at alpha 1 exactly.
	inline float scalar_lerp(float start, float end, float alpha) noexcept { return scalar_mul_add(end, alpha, scalar_neg_mul_sub(start, alpha, start)); }
	inline bool scalar_is_finite(float x) noexcept { return std::isfinite(x); }
	inline bool scalar_near_equal(float a, float b, float threshold = 0.00001F) noexcept { return std::fabs(a - b) < threshold; }
	inline bool scalar_greater_equal(float a, float b) noexcept { return a >= b; }
	inline bool scalar_greater_than(float a, float b) noexcept { return a > b; }
	// Round half away from zero.
	inline float scalar_round_symmetric(float x) noexcept { return x >= 0.0F ? std::floor(x + 0.5F) : std::ceil(x - 0.5F); }
	// Round half to even.
	inline float scalar_round_bankers(float x) noexcept { return std::nearbyint(x); }

	// scalarf (value kept in an SSE register) flavours used by the reference's scalar-track decoder
	inline scalarf scalar_mul_add(scalarf a, scalarf b, scalarf c) noexcept { return scalarf{ _mm_add_ss(_mm_mul_ss(a.value, b.value), c.value) }; }
	inline scalarf scalar_neg_mul_sub(scalarf a, scalarf b, scalarf c) noexcept { return scalarf{ _mm_sub_ss(c.value, _mm_mul_ss(a.value, b.value)) }; }
	inline scalarf scalar_lerp(scalarf start, scalarf end, scalarf alpha) noexcept { return scalar_mul_add(end, alpha, scalar_neg_mul_sub(start, alpha, start)); }

	inline float scalar_safe_to_float(int32_t x) noexcept { return static_cast<float>(x); }
	inline float scalar_safe_to_float(uint32_t x) noexcept { return static_cast<float>(x); }
	inline float scalar_safe_to_float(int64_t x) noexcept { return static_cast<float>(x); }
	inline float scalar_safe_to_float(uint64_t x) noexcept { return static_cast<float>(x); }
	inline float scalar_safe_to_float(float x) noexcept { return x; }
	inline float scalar_safe_to_float(double x) noexcept { return static_cast<float>(x); }
}
