#!/usr/bin/env python3
"""Mutated clips that the validators ACCEPT, registered and decoded on the GPU: what aclhip_check_clip lets through must decode without
reading outside the clip's memory (a GPU memory fault ends the process: the run says which mutation it was working on) and, since the
oracle walks the same bytes, to the oracle's bits. The mutations are the ones of tools/fuzz_host_validators.py (header words to
extremes, byte flips in headers and data, swapped words; no truncations -- a registered blob is whole).
usage: fuzz_gpu_mutated.py [seed] [seconds]"""
import faulthandler
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402,F401  (first: see tests/conftest.py)
from acl_amd import runtime, synth  # noqa: E402
from oracle import bindings as ob  # noqa: E402

EXTREMES = [0, 1, 2, 3, 0xFFFFFFFF, 0x7FFFFFFF, 0x80000000, 0xFFFF, 0x10000, 31, 32, 33]


def mutate(rng, blob):
    size = blob.size
    m = synth.aligned_bytes(size)
    m[:] = blob
    kind = rng.integers(0, 5)
    if kind == 0:       # byte flips in the data (bit rates, ranges, keyframes)
        for _ in range(int(rng.integers(1, 12))):
            m[rng.integers(100, size)] = rng.integers(0, 256)
    elif kind == 1:     # byte flips in the headers
        for _ in range(int(rng.integers(1, 4))):
            m[rng.integers(8, min(size, 200))] = rng.integers(0, 256)
    elif kind == 2:     # a header word to an extreme / near the size
        offset = int(rng.integers(2, min(size, 200) // 4)) * 4
        m[offset:offset + 4].view(np.uint32)[0] = int(rng.choice(EXTREMES + [size, size - 4, size // 2]))
    elif kind == 3:     # a small change of a header word (counts and offsets that stay plausible)
        offset = int(rng.integers(2, min(size, 200) // 4)) * 4
        word = m[offset:offset + 4].view(np.uint32)
        word[0] = np.uint32((int(word[0]) + int(rng.choice([-8, -4, -1, 1, 4, 8, 16]))) & 0xFFFFFFFF)
    else:               # two header words swapped
        a, b = (int(v) * 4 for v in rng.integers(2, min(size, 200) // 4, size=2))
        word = m[a:a + 4].copy()
        m[a:a + 4] = m[b:b + 4]
        m[b:b + 4] = word
    return m


def main():
    faulthandler.enable()       # a host side crash names the call it was in
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    seconds = float(sys.argv[2]) if len(sys.argv) > 2 else 60.0
    rng = np.random.default_rng(seed)
    import conftest
    sources = [synth.build_clip(**spec) for spec in conftest.CLIP_SPECS.values() if spec.get("num_tracks", 100) <= 330]
    # ... and what the reference's own compressor writes in the full-precision and mixed formats (round 6: quatf_full, quatf_drop_w_full,
    # vector3f_full; one segment of any length when nothing is variable), with and without optional metadata
    import helpers
    import types
    sources += [types.SimpleNamespace(blob=clip["blob"]) for clip in helpers.load_corpus()
                if (clip["spec"]["config"] in ("raw", "mixed_var_0", "mixed_var_1", "drop_w_full") or clip["name"].split("_", 1)[1].startswith("metadata")) and clip["spec"]["bones"] <= 105]
    if os.environ.get("FUZZ_SCALAR", "0") == "1":
        sources = [synth.build_scalar_clip(**spec) for spec in helpers.SCALAR_CLIP_SPECS.values()]
    accepted = refused = decoded = different = 0
    start = time.time()
    with runtime.Context(0) as context:
        while time.time() - start < seconds:
            source = sources[rng.integers(0, len(sources))]
            blob = mutate(rng, source.blob)
            status, _ = runtime.check_clip(blob, check_hash=False)
            if status != 0:
                refused += 1
                continue
            accepted += 1
            print(f"mutation {accepted + refused} (seed {seed}): registering", flush=True) if os.environ.get("FUZZ_VERBOSE") else None
            try:
                handle = context.register_clip(blob, check_hash=False)
            except runtime.AclHipError:
                refused += 1
                continue
            info = context.clip_info(handle)
            if info.num_components != 12:
                # a scalar track list (float1f .. vector4f)
                duration = float(info.duration) if np.isfinite(info.duration) else 1.0
                times = np.nan_to_num(np.concatenate([rng.uniform(-0.1, max(duration, 0.0) + 0.1, size=12), [0.0, duration]]).astype(np.float32), nan=0.0, posinf=1.0, neginf=0.0)
                if info.num_tracks != 0 and info.num_tracks < 100000:
                    values = context.decompress_scalar_tracks(np.full(times.size, handle, dtype=np.uint32), times)
                    decoded += times.size
                    row_floats = values.shape[1] * values.shape[2]
                    expected = ob.oracle_scalar_decompress_tracks_batch([blob], np.zeros(times.size, dtype=np.uint32), times, row_floats).reshape(values.shape)
                    if not np.array_equal(np.nan_to_num(values), np.nan_to_num(expected)):
                        different += 1
                        print(f"DIFFERENT (scalar): mutation {accepted + refused} of seed {seed}, {info.num_tracks} tracks x {info.num_components}", flush=True)
                        if different <= 12 and os.environ.get("FUZZ_SAVE_DIR"):
                            np.savez(os.path.join(os.environ["FUZZ_SAVE_DIR"], f"different_scalar_{seed}_{accepted + refused}.npz"), blob=blob, original=source.blob, times=times, gpu=values, oracle=expected)
                context.unregister_clip(handle)
                continue
            duration = float(info.duration) if np.isfinite(info.duration) else 1.0
            times = np.concatenate([rng.uniform(-0.1, max(duration, 0.0) + 0.1, size=12), [0.0, duration]]).astype(np.float32)
            times = np.nan_to_num(times, nan=0.0, posinf=1.0, neginf=0.0)
            poses = context.decompress_tracks(np.full(times.size, handle, dtype=np.uint32), times, num_tracks=max(info.num_tracks, 1))
            decoded += times.size
            if info.num_tracks != 0:
                expected = ob.oracle_decompress_tracks_batch([blob], np.zeros(times.size, dtype=np.uint32), times, info.num_tracks)
                same = np.array_equal(poses.view(np.uint32)[:, : info.num_tracks], expected.view(np.uint32))
                # (NaN payloads aside: a mutated range or raw sample may be a NaN, whose sign the two arithmetic units propagate differently: DESIGN 4.4)
                if not same and not np.array_equal(np.nan_to_num(poses[:, : info.num_tracks]), np.nan_to_num(expected)):
                    different += 1
                    print(f"DIFFERENT: mutation {accepted + refused} of seed {seed}, clip of {info.num_tracks} tracks", flush=True)
                    if different <= 12 and os.environ.get("FUZZ_SAVE_DIR"):
                        np.savez(os.path.join(os.environ["FUZZ_SAVE_DIR"], f"different_{seed}_{accepted + refused}.npz"), blob=blob, original=source.blob, times=times, gpu=poses[:, : info.num_tracks], oracle=expected)
            context.unregister_clip(handle)
        rejected = context.rejected_instance_count()
    print(f"gpu mutated fuzz {'ok' if different == 0 else 'FAILED'}: {accepted} accepted, {refused} refused, {decoded} poses decoded, {different} clips differ from the oracle, {rejected} instances refused by the kernels")
    sys.exit(0 if different == 0 else 1)


if __name__ == "__main__":
    main()
