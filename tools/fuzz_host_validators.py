#!/usr/bin/env python3
"""Mutated blobs against the HOST ONLY entry points -- aclhip_check_clip, aclhip_analyze_clip, aclhip_check_database,
aclhip_strip_database_tier, aclhip_plan_hierarchy_walk -- everything a registration does before it touches the device: a mutated
buffer is accepted or refused, never read out of bounds and never sized into a giant allocation. Meant to run under AddressSanitizer:

    hipcc --offload-arch=gfx950 -O1 -g -std=c++17 -ffp-contract=off -fPIC -shared -ldl -fsanitize=address -fno-omit-frame-pointer \
        acl_amd/csrc/aclhip.hip -o /tmp/libaclhip_asan.so
    ASAN_OPTIONS=detect_leaks=0 LD_PRELOAD=$(hipcc --print-file-name=libclang_rt.asan-x86_64.so) ACLHIP_LIBRARY=/tmp/libaclhip_asan.so \
        python tools/fuzz_host_validators.py [seed] [seconds]

(no GPU needed; without the sanitizer it still finds crashes and runaway allocations). Round 5: 0xFFFFFFFF tracks wrapped the
sub-track type word count to zero, passed validation and sized the derived tables for 4 G tracks -- found in the first minute."""
import ctypes
import os
import resource
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from acl_amd import runtime, synth  # noqa: E402

EXTREMES = [0, 1, 2, 0xFFFFFFFF, 0xFFFFFFFE, 0x7FFFFFFF, 0x80000000, 0xFFFF, 0x10000, 0xFFFFFFF0, 0xFFFFFFF1]


def mutate(rng, blob, header_bytes):
    """one mutated copy (16 byte aligned, 64 bytes of slack behind it) and the size to claim for it"""
    size = blob.size
    m = synth.aligned_bytes(size + 64)
    m[:size] = blob
    kind = rng.integers(0, 6)
    if kind == 0:       # byte flips anywhere
        for _ in range(int(rng.integers(1, 8))):
            m[rng.integers(0, size)] = rng.integers(0, 256)
    elif kind == 1:     # byte flips in the headers
        for _ in range(int(rng.integers(1, 6))):
            m[rng.integers(0, min(size, header_bytes))] = rng.integers(0, 256)
    elif kind == 2:     # a header word set to an extreme, or to something near the buffer's size
        offset = int(rng.integers(0, min(size, header_bytes) // 4)) * 4
        m[offset:offset + 4].view(np.uint32)[0] = int(rng.choice(EXTREMES + [size, size - 1, size + 1, size - 4, size // 2]))
    elif kind == 3:     # truncation
        size = int(rng.integers(0, size))
    elif kind == 4:     # the size field lies
        m[0:4].view(np.uint32)[0] = int(rng.integers(0, 2 * size))
    else:               # two header words swapped
        a, b = (int(v) * 4 for v in rng.integers(0, min(size, header_bytes) // 4, size=2))
        word = m[a:a + 4].copy()
        m[a:a + 4] = m[b:b + 4]
        m[b:b + 4] = word
    return m, size


def main():
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    seconds = float(sys.argv[2]) if len(sys.argv) > 2 else 60.0
    # a validator that sizes a table from an untrusted count shows up as a failed allocation instead of an hour of swapping
    resource.setrlimit(resource.RLIMIT_AS, (64 << 30, 64 << 30)) if "asan" not in os.environ.get("ACLHIP_LIBRARY", "") else None
    rng = np.random.default_rng(seed)
    import conftest
    import helpers
    clips = [synth.build_clip(**spec).blob for spec in conftest.CLIP_SPECS.values() if spec.get("num_tracks", 100) <= 400]
    clips += [synth.build_scalar_clip(seed=3 + track_type, track_type=track_type, num_tracks=17, num_samples=23).blob for track_type in range(5)]
    databases = []
    for name in helpers.database_golden_cases():
        case = helpers.load_database_golden(name)
        databases.append(case)
        clips += [c for c in case["clips"]][:2]
    lib = runtime.load_library()
    facts = ctypes.c_uint32(0)
    message = ctypes.create_string_buffer(256)
    counts = {"clip ok": 0, "clip refused": 0, "database ok": 0, "database refused": 0, "strip ok": 0, "strip refused": 0, "walk ok": 0, "walk refused": 0}
    start = time.time()
    while time.time() - start < seconds:
        which = rng.integers(0, 10)
        if which < 6:
            m, size = mutate(rng, clips[rng.integers(0, len(clips))], 200)
            status = lib.aclhip_check_clip(m.ctypes.data, size, 0, message, 256)
            analysis = lib.aclhip_analyze_clip(m.ctypes.data, size, 0, ctypes.byref(facts))
            assert (status == 0) == (analysis == 0), (status, analysis)
            counts["clip ok" if status == 0 else "clip refused"] += 1
        elif which < 9 and databases:
            # (the INLINE form: bulk data inside the buffer, whose size the call knows. Split bulk data arrives as bare pointers, like in the
            # reference's database_context::initialize: its size is whatever the header says -- a contract, nothing a validator can check)
            case = databases[rng.integers(0, len(databases))]
            m, size = mutate(rng, case["database_inline"], 256)
            status = lib.aclhip_check_database(m.ctypes.data, size, None, None, 0, message, 256)
            counts["database ok" if status == 0 else "database refused"] += 1
            out_size = ctypes.c_uint64(0)
            tier = int(rng.integers(0, 3))
            status = lib.aclhip_strip_database_tier(m.ctypes.data, size, tier, None, 0, ctypes.byref(out_size))
            if status == 0 and out_size.value < (1 << 28):
                out = synth.aligned_bytes(out_size.value + 16)
                status = lib.aclhip_strip_database_tier(m.ctypes.data, size, tier, out.ctypes.data, out_size.value, ctypes.byref(out_size))
            counts["strip ok" if status == 0 else "strip refused"] += 1
        elif which == 9 and rng.integers(0, 2) == 0:
            # the host side decode order: any clip handles, any number of wavefronts per pose
            n = int(rng.integers(0, 3000))
            handles = rng.integers(0, int(rng.choice([1, 7, 300, 0xFFFFFFFF])) + 1, size=n, dtype=np.uint64).astype(np.uint32)
            order = np.zeros(max(n, 1), dtype=np.uint32)
            status = lib.aclhip_order_instances_for_pose_windows(int(rng.choice([0, 1, 2, 3, 8, 17, 0xFFFFFFFF])), handles.ctypes.data, n, order.ctypes.data)
            assert status != 0 or np.array_equal(np.sort(order[:n]), np.arange(n)), "not a permutation"
            counts["order ok" if status == 0 else "order refused"] = counts.get("order ok" if status == 0 else "order refused", 0) + 1
        else:
            n = int(rng.integers(1, 300))
            parents = np.array([0xFFFFFFFF if i == 0 else rng.integers(0, i) for i in range(n)], dtype=np.uint32)
            for _ in range(int(rng.integers(0, 4))):
                parents[rng.integers(0, n)] = int(rng.choice(EXTREMES + [n, n - 1, n + 1]))
            steps = np.zeros(n, dtype=np.uint32)
            num_steps = ctypes.c_uint32(0)
            status = lib.aclhip_plan_hierarchy_walk(parents.ctypes.data, n, int(rng.choice([1, 2, 64, 0, 0xFFFFFFFF])), steps.ctypes.data, ctypes.byref(num_steps))
            counts["walk ok" if status == 0 else "walk refused"] += 1
    print("host validator fuzz ok:", ", ".join(f"{value} {key}" for key, value in counts.items()))


if __name__ == "__main__":
    main()
