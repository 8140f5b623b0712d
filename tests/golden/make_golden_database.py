#!/usr/bin/env python3
"""Generates tests/golden/database/*.npz: clips split by the REFERENCE's build_database() into a compressed_database with
medium / low importance tiers, a script of stream_in / stream_out requests, and the poses the reference's own
decompression_context<..>::initialize(tracks, database_context) + seek + decompress_tracks produces after every request
(oracle/_ref/libaclref_db.so, built from /root/reference by oracle/Makefile from oracle/ref_database_bridge.cpp).

Run in the build container (where /root/reference exists):  python tests/golden/make_golden_database.py
Stored per case: clip blobs (concatenated, with offsets), the database with its bulk data split off (+ both bulk buffers) and the
same database with inline bulk data, ops [n, 3] = (tier, num_chunks, 1 stream_in / 0 stream_out), times [num_clips, T],
policies, poses [1 + n, num_clips, len(policies), T, max_tracks, 12] (state 0 = nothing streamed in).
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from acl_amd import synth  # noqa: E402
from oracle import bindings as ob  # noqa: E402

ALL = 0xFFFFFFFF
CASES = {
    # name: (raw clips [(seed, tracks, samples)], build options, ops)
    "three_clips_4k_chunks": (
        [(60, 30, 150), (61, 20, 90), (62, 25, 200)], dict(medium_proportion=0.3, low_proportion=0.4, max_chunk_size=4096),
        [(1, 1, 1), (2, 2, 1), (1, 1, 0), (1, 2, 1), (1, ALL, 1), (2, 1, 0), (2, ALL, 1), (1, ALL, 0), (2, ALL, 0), (2, ALL, 1), (1, ALL, 1)]),
    "two_clips_single_chunk": (
        [(70, 16, 64), (71, 28, 40)], dict(medium_proportion=0.25, low_proportion=0.25, max_chunk_size=64 * 1024),
        [(2, ALL, 1), (1, ALL, 1), (2, ALL, 0), (1, ALL, 0)]),
    # num_chunks = 0: the reference's `first + 0 - 1` wraps when the first candidate chunk is chunk 0 and the WHOLE tier moves
    # (database.impl.h:490-492,571-573); with any other first chunk nothing does. (stream_out(tier, 0) while later chunks are NOT resident
    # walks chunk headers that were never streamed in and crashes the reference: not part of the script.)
    "zero_chunk_requests": (
        [(90, 60, 300), (91, 40, 220)], dict(medium_proportion=0.3, low_proportion=0.3, max_chunk_size=4096),
        [(2, 0, 1), (2, 0, 0), (1, 1, 1), (1, 0, 1), (1, ALL, 1), (1, 0, 0), (2, 2, 1), (2, 0, 1), (2, ALL, 1), (1, ALL, 1)]),
    "medium_tier_only": (
        [(80, 22, 120), (81, 12, 33)], dict(medium_proportion=0.5, low_proportion=0.0, max_chunk_size=4096),
        [(1, 1, 1), (2, ALL, 1), (1, ALL, 1), (1, 1, 0)]),
}
POLICIES = [0, 3]     # none, nearest
NUM_TIMES = 8


def main():
    if not ob.have_ref_database():
        raise SystemExit("oracle/_ref/libaclref_db.so is missing: run `make -C oracle ref` where /root/reference exists")
    rng = np.random.default_rng(777)
    os.makedirs(os.path.join(HERE, "database"), exist_ok=True)
    only = sys.argv[1:]
    for name, (raw_specs, build_options, ops) in CASES.items():
        if only and name not in only:
            continue
        blobs = []
        for seed, num_tracks, num_samples in raw_specs:
            raw_clip = synth.build_clip(seed=seed, num_tracks=num_tracks, num_samples=num_samples, with_side_data=True)
            blobs.append(ob.ref_db_compress(raw_clip.raw_keyframes, raw_clip.sample_rate))
        ref = ob.ReferenceDatabase(blobs, **build_options)
        max_tracks = max(spec[1] for spec in raw_specs)
        times = np.zeros((len(blobs), NUM_TIMES), dtype=np.float32)
        for c, clip in enumerate(ref.clips):
            duration = ob.ref().aclref_get_duration(clip.ctypes.data, -1)
            times[c] = np.concatenate([rng.uniform(0.0, duration, size=NUM_TIMES - 2), [0.0, duration]])

        poses = np.zeros((1 + len(ops), len(blobs), len(POLICIES), NUM_TIMES, max_tracks, 12), dtype=np.float32)

        def snapshot(state):
            for c in range(len(blobs)):
                for p, policy in enumerate(POLICIES):
                    for i, t in enumerate(times[c]):
                        out = ref.decompress(c, float(t), policy)
                        out[:, 7] = 0.0         # W lanes of translation / scale are unspecified in the reference
                        out[:, 11] = 0.0
                        poses[state, c, p, i, : out.shape[0]] = out

        snapshot(0)
        results = []
        for state, (tier, num_chunks, stream_in) in enumerate(ops):
            results.append(ref.stream(tier, num_chunks, bool(stream_in)))
            snapshot(state + 1)

        # strip_database_quality_tier of the inline and of the split database, both tiers (empty array: the reference refuses)
        stripped = {}
        for split in (0, 1):
            for tier in (1, 2):
                blob = ref.strip(tier, bool(split))
                stripped[f"stripped_{'split' if split else 'inline'}_{'medium' if tier == 1 else 'low'}"] = np.asarray(blob) if blob is not None else np.zeros(0, np.uint8)

        offsets = np.cumsum([0] + [clip.size for clip in ref.clips]).astype(np.int64)
        path = os.path.join(HERE, "database", f"{name}.npz")
        np.savez_compressed(path, clips=np.concatenate(ref.clips), clip_offsets=offsets, database=np.asarray(ref.database),
                            database_inline=np.asarray(ref.database_inline), bulk_medium=np.asarray(ref.bulk[1]), bulk_low=np.asarray(ref.bulk[2]),
                            ops=np.array(ops, dtype=np.uint32), results=np.array(results, dtype=np.int32), times=times,
                            policies=np.array(POLICIES, dtype=np.uint8), poses=poses, **stripped)
        print(f"{name}: {os.path.getsize(path)} bytes, chunks {ref.num_chunks}, results {results}")
        ref.close()


if __name__ == "__main__":
    main()
