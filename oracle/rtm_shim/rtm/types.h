#pragma once
// TEST INFRASTRUCTURE (oracle/): Realtime Math stand-in, x86 SSE2 flavour only. See impl/detect_compiler.h.
#include "rtm/math.h"
#include "rtm/impl/compiler_utils.h"
#include "rtm/impl/bit_cast.impl.h"

#if !defined(RTM_SSE2_INTRINSICS)
	#error "oracle/rtm_shim only implements the SSE2 flavour of Realtime Math"
#endif

namespace rtm
{
	using vector4f = __m128;
	using quatf = __m128;
	using mask4f = __m128;

	// A float kept in lane 0 of an SSE register; converts to and from float like RTM's scalarf does through scalar_cast / scalar_set.
	struct scalarf
	{
		__m128 value;
		scalarf() = default;
		scalarf(__m128 value_) noexcept : value(value_) {}
		scalarf(float value_) noexcept : value(_mm_set_ps1(value_)) {}
		operator float() const noexcept { return _mm_cvtss_f32(value); }
	};

	struct float2f { float x; float y; };
	struct float3f { float x; float y; float z; };
	struct alignas(4) float4f { float x; float y; float z; float w; };

	struct qvvf
	{
		quatf rotation;
		vector4f translation;
		vector4f scale;
	};

	// Three orthonormal-ish axes plus a translation row, each a vector4f (W ignored).
	struct matrix3x4f
	{
		vector4f x_axis;
		vector4f y_axis;
		vector4f z_axis;
		vector4f w_axis;
	};

	enum class mix4 { x = 0, y = 1, z = 2, w = 3, a = 4, b = 5, c = 6, d = 7 };
	enum class axis4 { x = 0, y = 1, z = 2, w = 3 };

	using vector4f_arg0 = const vector4f;
	using vector4f_arg1 = const vector4f;
	using vector4f_arg2 = const vector4f;
	using vector4f_arg3 = const vector4f;
	using vector4f_arg4 = const vector4f;
	using vector4f_arg5 = const vector4f;
	using vector4f_arg6 = const vector4f;
	using vector4f_arg7 = const vector4f;
	using vector4f_argn = const vector4f&;

	using quatf_arg0 = const quatf;
	using quatf_arg1 = const quatf;
	using quatf_arg2 = const quatf;
	using quatf_argn = const quatf&;

	using mask4f_arg0 = const mask4f;
	using mask4f_arg1 = const mask4f;
	using mask4f_argn = const mask4f&;

	using scalarf_arg0 = const scalarf;
	using scalarf_arg1 = const scalarf;
	using scalarf_argn = const scalarf&;

	using qvvf_arg0 = const qvvf&;
	using qvvf_arg1 = const qvvf&;
	using qvvf_argn = const qvvf&;

	using matrix3x4f_arg0 = const matrix3x4f&;
	using matrix3x4f_arg1 = const matrix3x4f&;
	using matrix3x4f_argn = const matrix3x4f&;
}
