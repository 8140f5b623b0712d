"""debug aid: hierarchy replaced under launches in flight (tests/test_gpu_lifetime.py), checked after every round"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from acl_amd import runtime, synth
from oracle import bindings as ob
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import helpers

clip = synth.build_clip(seed=2, num_tracks=100, num_samples=301, sample_rate=30.0)
device = torch.device("cuda", 0)
n = 4096
rng = np.random.default_rng(3)
times = rng.uniform(0.0, clip.duration, size=n).astype(np.float32)
hierarchies = [synth.humanoid_hierarchy(100), np.concatenate([[runtime.NO_PARENT], np.arange(99)]).astype(np.uint32)]
expected = [ob.oracle_decompress_poses_batch([clip.blob], np.zeros(n, dtype=np.uint32), times, 100, parent_indices=h) for h in hierarchies]
for sync_between in (True, False):
    with runtime.Context(0) as context:
        handle = context.register_clip(clip.blob)
        d_clips = torch.full((n,), handle, dtype=torch.int32, device=device)
        d_times = torch.from_numpy(times).to(device)
        d_poses = torch.zeros((n, 100, 12), dtype=torch.float32, device=device)
        consumers = runtime.PoseConsumers()
        consumers.object_space = 1
        stream = torch.cuda.Stream(device)
        for round_index in range(6):
            which = round_index % 2
            context.set_clip_hierarchy(handle, hierarchies[which])
            for _ in range(4):
                context.decompress_poses_batch(d_clips.data_ptr(), d_times.data_ptr(), n, d_poses.data_ptr(), 4800, consumers, stream=stream.cuda_stream)
            if sync_between:
                stream.synchronize()
                poses = d_poses.cpu().numpy()
                bad = np.flatnonzero(np.any(poses.view(np.uint32)[..., helpers.XYZ_LANES] != expected[which].view(np.uint32)[..., helpers.XYZ_LANES], axis=(1, 2)))
                print("sync", round_index, which, "bad instances", bad.size, bad[:8], flush=True)
                if bad.size:
                    i = bad[0]
                    tracks = np.flatnonzero(np.any(poses[i].view(np.uint32)[:, helpers.XYZ_LANES] != expected[which][i].view(np.uint32)[:, helpers.XYZ_LANES], axis=1))
                    print("  instance", i, "bad tracks", tracks[:20], "max diff", np.abs(poses[i] - expected[which][i]).max())
        stream.synchronize()
        poses = d_poses.cpu().numpy()
        bad = np.flatnonzero(np.any(poses.view(np.uint32)[..., helpers.XYZ_LANES] != expected[1].view(np.uint32)[..., helpers.XYZ_LANES], axis=(1, 2)))
        print("final (sync_between=%s): bad instances" % sync_between, bad.size, bad[:8], "rejected", context.rejected_instance_count(), flush=True)
