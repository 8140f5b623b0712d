#pragma once
// Architecture feature gates, as rtm/math.h derives them from the compiler's predefined macros.
#include "rtm/impl/detect_compiler.h"
#include "rtm/impl/detect_cpp_version.h"
#if defined(__x86_64__) || defined(__i386__)
	#define RTM_ARCH_X64
#endif
#if !defined(RTM_NO_INTRINSICS)
	#if defined(__SSE2__)
		#define RTM_SSE2_INTRINSICS
	#endif
	#if defined(__SSE3__)
		#define RTM_SSE3_INTRINSICS
	#endif
	#if defined(__SSE4_1__)
		#define RTM_SSE4_INTRINSICS
	#endif
	#if defined(__AVX__)
		#define RTM_AVX_INTRINSICS
	#endif
	#if defined(__AVX2__)
		#define RTM_AVX2_INTRINSICS
	#endif
#endif
#if defined(RTM_SSE2_INTRINSICS)
	#include <immintrin.h>
#endif
#include <cstdint>
#include <cmath>
