#pragma once
#include "rtm/impl/detect_compiler.h"
#include "rtm/impl/detect_cpp_version.h"
#define RTM_FORCE_INLINE __attribute__((always_inline)) inline
#define RTM_FORCE_NOINLINE __attribute__((noinline))
#define RTM_SIMD_CALL
#define RTM_NO_EXCEPT noexcept
#define RTM_DISABLE_SECURITY_COOKIE_CHECK
#define RTM_IMPL_FILE_PRAGMA_PUSH
#define RTM_IMPL_FILE_PRAGMA_POP
#define RTM_DEPRECATED(msg) [[deprecated(msg)]]
