#!/usr/bin/env python3
"""Generates tests/golden/bench/database_64_clips_100_bones.npz: the input of `bench.py --workload database` and of the full-size
database parity test (BASELINE.json configs[4] as SURVEY 8(d)5 writes it): 64 clips of 100 bones compressed with database support and split by the
REFERENCE's build_database() with its default tier proportions (medium 0 %, low 50 %, compression_settings.h:64-69) into 64 KiB
chunks. No poses are stored: parity of the database path is covered by the other fixtures of this directory.

Run in the build container (where /root/reference exists):  python tests/golden/make_bench_database.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from acl_amd import synth  # noqa: E402
from oracle import bindings as ob  # noqa: E402


NUM_CLIPS = 64


def main():
    if not ob.have_ref_database():
        raise SystemExit("oracle/_ref/libaclref_db.so is missing: run `make -C oracle ref` where /root/reference exists")
    rng = np.random.default_rng(5)
    blobs = []
    for index in range(NUM_CLIPS):
        num_samples = int(rng.integers(60, 240))
        raw_clip = synth.build_clip(seed=500 + index, num_tracks=100, num_samples=num_samples, with_side_data=True)
        blobs.append(ob.ref_db_compress(raw_clip.raw_keyframes, raw_clip.sample_rate))
    reference = ob.ReferenceDatabase(blobs, medium_proportion=0.0, low_proportion=0.5, max_chunk_size=64 * 1024)
    offsets = np.cumsum([0] + [clip.size for clip in reference.clips]).astype(np.int64)
    path = os.path.join(HERE, "bench", f"database_{NUM_CLIPS}_clips_100_bones.npz")
    np.savez_compressed(path, clips=np.concatenate(reference.clips), clip_offsets=offsets, database=np.asarray(reference.database),
                        database_inline=np.zeros(0, np.uint8), bulk_medium=np.asarray(reference.bulk[1]), bulk_low=np.asarray(reference.bulk[2]))
    print(f"{path}: {os.path.getsize(path)} bytes, clips {offsets[-1]} bytes, bulk {reference.bulk[1].size} + {reference.bulk[2].size} bytes, chunks {reference.num_chunks}")
    reference.close()


if __name__ == "__main__":
    main()
