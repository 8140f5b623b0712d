"""acl_amd -- MI355X-native batched decompression of ACL (nfrechette/acl) animation clips.

The product is the C ABI in include/aclhip.h implemented by acl_amd/csrc (HIP kernels for gfx950, built into
acl_amd/lib/libaclhip.so) and the C++ mirror of acl::decompression_context in acl_amd/csrc/aclhip.hpp.
This Python package is thin plumbing over that ABI for tests, benchmarks and torch interop:

    acl_amd.runtime   ctypes binding of libaclhip.so (Context, DecompressParams)
    acl_amd.synth     synthetic compressed_tracks writer (host only)
    acl_amd.build     in-tree native builds

There is no CPU fallback: importing acl_amd.runtime without the built HIP library raises.
"""
__all__ = ["build", "runtime", "synth"]
__version__ = "0.1.0"
