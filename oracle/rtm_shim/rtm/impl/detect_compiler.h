#pragma once
// TEST INFRASTRUCTURE (oracle/): minimal stand-in for Realtime Math (nfrechette/rtm v2.3.0, the
// un-vendored submodule named in /root/reference/external/README.md:7). It exists ONLY so that the
// reference's own, unmodified headers under /root/reference/includes can be compiled into
// oracle/_ref/. Nothing in the product (acl_amd/, include/) includes it.
#if defined(__clang__)
	#define RTM_COMPILER_CLANG
#elif defined(__GNUC__)
	#define RTM_COMPILER_GCC
#endif
