import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from acl_amd import runtime, synth
from oracle import bindings as ob
from conftest import CLIP_SPECS
import test_gpu_layouts as T
ctx = runtime.Context(0)
for name in sys.argv[1:]:
    clip = synth.build_clip(**CLIP_SPECS[name])
    handle = ctx.register_clip(clip.blob)
    rng = np.random.default_rng(len(name)); n = 257
    times = rng.uniform(-0.1, clip.duration + 0.1, size=n).astype(np.float32)
    oracle = ob.oracle_decompress_tracks_batch([clip.blob], np.zeros(n, dtype=np.uint32), times, clip.num_tracks)
    handles = np.full(n, handle, dtype=np.uint32)
    for layout in ("qvv48", "qvv40", "qv32"):
        for skip in ((0, 0, 0), (0, 0, 1), (1, 0, 0)):
            got = T.launch(ctx, handles, times, layout, skip, max_tracks=clip.num_tracks)
            exp = T.expected_through_layout(oracle, layout, skip)
            bad = np.argwhere(got.view(np.uint32) != exp.view(np.uint32))
            print(name, layout, skip, "mismatches", len(bad))
            if len(bad):
                print(" instances", np.unique(bad[:, 0])[:10], "tracks", np.unique(bad[:, 1])[:40], "components", np.unique(bad[:, 2]))
                i, t, c = bad[0]
                print(" first", i, t, c, got[i, t], exp[i, t])
    ctx.unregister_clip(handle)
