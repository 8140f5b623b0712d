#pragma once
// TEST INFRASTRUCTURE (oracle/): Realtime Math stand-in. Integer masks share the float mask representation here.
#include "rtm/vector4f.h"
namespace rtm
{
	using mask4i = __m128;
	inline mask4i mask_set(uint32_t x, uint32_t y, uint32_t z, uint32_t w) noexcept { return _mm_castsi128_ps(_mm_set_epi32(int32_t(w), int32_t(z), int32_t(y), int32_t(x))); }
	inline uint32_t mask_get_x(mask4i m) noexcept { return uint32_t(_mm_cvtsi128_si32(_mm_castps_si128(m))); }
	inline bool mask_all_true(mask4i m) noexcept { return _mm_movemask_ps(m) == 0xF; }
	inline bool mask_any_true(mask4i m) noexcept { return _mm_movemask_ps(m) != 0; }
}
