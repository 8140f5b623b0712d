"""Debugging aid: which instances / tracks of the grouped scalar kernel differ from the oracle (tests/test_gpu_scalar.py's spec 1)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from acl_amd import runtime, synth
from oracle import bindings as ob

spec = dict(seed=10, track_type=4, num_tracks=70, num_samples=60, raw_fraction=0.1)
clip = synth.build_scalar_clip(**spec)
n = 16384 + 3
rng = np.random.default_rng(10)
times = (rng.uniform(-0.05, 1.05, size=n).astype(np.float32) * np.float32(clip.duration)).astype(np.float32)
with runtime.Context(0) as context:
    handle = context.register_clip(clip.blob)
    values = context.decompress_scalar_tracks(np.full(n, handle, dtype=np.uint32), times)
    bad = 0
    for i in range(n):
        expected = ob.oracle_scalar_decompress_tracks(clip.blob, float(times[i]), 0, None)
        got = values[i, : clip.num_tracks]
        if not np.array_equal(got.view(np.uint32), expected.view(np.uint32)):
            wrong = np.argwhere(got.view(np.uint32) != expected.view(np.uint32))
            bad += 1
            if bad <= 12:
                print("instance", i, "time", times[i], "kf", times[i] * 30.0, "wrong entries", len(wrong), "first", wrong[:3].tolist(), got.reshape(-1)[:0])
                t, c = wrong[0]
                print("   got", got[t], "expected", expected[t])
    print("bad instances", bad, "of", n)
