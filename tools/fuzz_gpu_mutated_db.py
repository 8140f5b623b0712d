#!/usr/bin/env python3
"""Mutated compressed_database headers and bulk data that the validators ACCEPT, bound, streamed and decoded on the GPU (the database
side of tools/fuzz_gpu_mutated.py; DESIGN.md 9.6 of round 5): what aclhip_check_database + aclhip_register_clip_with_database let through
must stream in and out in any chunk counts and decode without reading outside the device buffers (a GPU memory fault ends the process: the
run says which mutation it was on) and to the bits of the restated database_context (oracle/database.py, which
tests/test_oracle_vs_reference.py holds to the reference's own database_context, decompression/database/impl/database.impl.h:443-640).
Sources: the four corpus databases (tests/golden/corpus) and the committed database fixtures. Mutations: header words to extremes, small
changes of counts and offsets, byte flips in the chunk descriptions / clip metadata, byte flips and word edits in the chunk headers and
chunk segment headers of the bulk data, swapped words.
usage: fuzz_gpu_mutated_db.py [seed] [seconds]"""
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
from oracle.database import OracleDatabase  # noqa: E402

EXTREMES = [0, 1, 2, 3, 0xFFFFFFFF, 0x7FFFFFFF, 0x80000000, 0xFFFF, 0x10000, 31, 32, 33, 4096, 4095]


def aligned_copy(array):
    out = synth.aligned_bytes(max(array.size, 1))
    out[: array.size] = array
    return out[: array.size] if array.size else out[:0]


def grown(buffer, size):
    """buffer, or a zero padded aligned copy of at least `size` bytes"""
    if buffer.size >= size:
        return buffer
    out = synth.aligned_bytes(size)
    out[:] = 0
    out[: buffer.size] = buffer
    return out


def mutate_words(rng, buffer, first, last):
    """one mutation of the 32 bit words in [first, last) of buffer (in place)"""
    if last - first < 8:
        return
    kind = rng.integers(0, 4)
    word_at = lambda: first + int(rng.integers(0, (last - first) // 4)) * 4
    if kind == 0:
        for _ in range(int(rng.integers(1, 4))):
            buffer[int(rng.integers(first, last))] = rng.integers(0, 256)
    elif kind == 1:
        offset = word_at()
        buffer[offset:offset + 4].view(np.uint32)[0] = int(rng.choice(EXTREMES + [buffer.size, buffer.size - 4, buffer.size // 2]))
    elif kind == 2:
        offset = word_at()
        word = buffer[offset:offset + 4].view(np.uint32)
        word[0] = np.uint32((int(word[0]) + int(rng.choice([-16, -8, -4, -1, 1, 4, 8, 16]))) & 0xFFFFFFFF)
    else:
        a, b = word_at(), word_at()
        saved = buffer[a:a + 4].copy()
        buffer[a:a + 4] = buffer[b:b + 4]
        buffer[b:b + 4] = saved


def mutate(rng, case):
    database, medium, low = aligned_copy(case["database"]), aligned_copy(case["bulk_medium"]), aligned_copy(case["bulk_low"])
    target = rng.integers(0, 3)
    if target == 0 or (medium.size == 0 and low.size == 0):
        mutate_words(rng, database, 8, database.size)                         # database_header, chunk descriptions, clip metadata
    else:
        bulk = medium if (target == 1 and medium.size) or low.size == 0 else low
        # chunk headers and chunk segment headers sit at the start of every chunk: aim at the first 256 bytes of a random 4 KiB page, or anywhere
        if rng.uniform() < 0.7:
            page = int(rng.integers(0, max(bulk.size // 4096, 1))) * 4096
            mutate_words(rng, bulk, page, min(page + 256, bulk.size))
        else:
            mutate_words(rng, bulk, 0, bulk.size)
    return database, medium, low


def main():
    faulthandler.enable()       # a host side crash names the call it was in
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    seconds = float(sys.argv[2]) if len(sys.argv) > 2 else 60.0
    rng = np.random.default_rng(seed)
    import helpers
    cases = [helpers.load_corpus_database(name) for name in helpers.CORPUS_DATABASES] + [helpers.load_database_golden(name) for name in helpers.database_golden_cases()]
    accepted = refused = unbound = decoded = different = streamed = 0
    start = time.time()
    with runtime.Context(0) as context:
        iteration = 0
        while time.time() - start < seconds:
            iteration += 1
            case = cases[int(rng.integers(0, len(cases)))]
            database, medium, low = mutate(rng, case)
            # The caller's side of the contract: the bulk data buffers hold database_header::bulk_data_size bytes (the entry points take
            # pointers without sizes, like database_context::initialize with its streamers). A mutated size that claims more gets a buffer
            # that long (zeros behind the real data) while that is reasonable, and is not a case otherwise.
            claimed = [int(v) for v in database[40:48].view(np.uint32)] if database.size >= 48 else [0, 0]
            if max(claimed) > (16 << 20):
                refused += 1
                continue
            medium, low = (grown(buffer, size) for buffer, size in zip((medium, low), claimed))
            bulk = (medium if medium.size else None, low if low.size else None)
            status, _ = runtime.check_database(database, bulk[0], bulk[1], check_hash=False)
            if status != 0:
                refused += 1
                continue
            if os.environ.get("FUZZ_VERBOSE"):
                print(f"mutation {iteration} (seed {seed}): registering", flush=True)
            try:
                handle = context.register_database(database, bulk[0], bulk[1], check_hash=False)
            except runtime.AclHipError:
                refused += 1
                continue
            accepted += 1
            clips = []
            for blob in case["clips"]:
                try:
                    clips.append((blob, context.register_clip_with_database(blob, handle)))
                except runtime.AclHipError:
                    unbound += 1           # (the mutation took the clip out of the database, or made its keyframes leave the bulk data: refused at bind)
            try:
                oracle_db = OracleDatabase(database, medium, low)
            except Exception:               # noqa: BLE001 -- the restatement has no validation of its own: a header it cannot walk is not compared
                oracle_db = None
            info = context.database_info(handle)
            script = [(int(rng.integers(1, 3)), bool(rng.uniform() < 0.7), int(rng.choice([1, 2, 3, 0xFFFFFFFF]))) for _ in range(int(rng.integers(2, 7)))]
            for tier, stream_in, num_chunks in script:
                try:
                    moved = (context.database_stream_in if stream_in else context.database_stream_out)(handle, tier, num_chunks)
                except runtime.AclHipError:
                    break                   # (a chunk whose headers do not pass the checks made when it arrives: refused, nothing applied)
                streamed += moved
                if oracle_db is not None:
                    try:
                        expected_moved = (oracle_db.stream_in if stream_in else oracle_db.stream_out)(tier, num_chunks)
                    except Exception:       # noqa: BLE001
                        oracle_db = None
                    else:
                        if expected_moved != moved:
                            different += 1
                            print(f"DIFFERENT: mutation {iteration} of seed {seed}: {moved} chunks moved, the restated database_context moves {expected_moved}", flush=True)
                            oracle_db = None
                for blob, clip in clips:
                    clip_info = context.clip_info(clip)
                    duration = float(clip_info.duration)
                    times = np.concatenate([rng.uniform(0.0, max(duration, 0.0), size=6), [0.0, duration]]).astype(np.float32)
                    poses = context.decompress_tracks(np.full(times.size, clip, dtype=np.uint32), times)
                    decoded += times.size
                    if oracle_db is None:
                        continue
                    for row, t in enumerate(times):
                        expected = oracle_db.decompress_tracks(blob, float(t))
                        if not helpers.bit_equal(np.nan_to_num(poses[row]), np.nan_to_num(expected)):
                            different += 1
                            print(f"DIFFERENT: mutation {iteration} of seed {seed}: clip of {clip_info.num_tracks} tracks at t = {t}", flush=True)
                            if different <= 12 and os.environ.get("FUZZ_SAVE_DIR"):
                                np.savez(os.path.join(os.environ["FUZZ_SAVE_DIR"], f"different_db_{seed}_{iteration}.npz"), database=database, medium=medium, low=low, clip=blob, time=t, gpu=poses[row], oracle=expected)
                            break
            assert info.num_clips >= 0
            for _, clip in clips:
                context.unregister_clip(clip)
            context.unregister_database(handle)
        rejected = context.rejected_instance_count()
    print(f"gpu mutated database fuzz {'ok' if different == 0 else 'FAILED'}: {accepted} accepted, {refused} refused, {unbound} clips refused at bind, {streamed} chunks streamed, "
          f"{decoded} poses decoded, {different} differences, {rejected} instances refused by the kernels")
    sys.exit(0 if different == 0 else 1)


if __name__ == "__main__":
    main()
