"""Blend of K clip instances fused into the decode (aclhip_pose_consumers::num_blend_clips) through the C ABI: bit exact
against the CPU oracle's decode x K -> aclo_blend_poses -> apply_additive_to_base -> local_to_object_space pipeline (the definition is
pinned by an fp64 restatement in test_pose_consumers_oracle.py). Needs a GPU."""
import numpy as np
import pytest

from acl_amd import runtime, synth
from oracle import bindings as ob
import helpers

pytestmark = pytest.mark.gpu


def _hierarchy(rng, num_tracks):
    parents = np.zeros(num_tracks, dtype=np.uint32)
    parents[0] = runtime.NO_PARENT
    for i in range(1, num_tracks):
        parents[i] = rng.integers(max(0, i - 9), i)
    return parents


SHAPES = {
    "characters_100": dict(num_tracks=100, num_samples=61),
    "scaled_37": dict(num_tracks=37, num_samples=33, has_scale=1, scale_default=0.3),
    "two_windows_130": dict(num_tracks=130, num_samples=20, has_scale=1, scale_default=0.5),
    "stripped_wrap_50": dict(num_tracks=50, num_samples=100, strip_keyframes=1, wrap=1),
}


@pytest.mark.parametrize("name", sorted(SHAPES))
@pytest.mark.parametrize("num_blend", [2, 3, 4])
def test_blend_matches_the_oracle(name, num_blend):
    spec = SHAPES[name]
    rng = np.random.default_rng(1000 * num_blend + len(name))
    clips = [synth.build_clip(seed=700 + k, **dict(spec, num_samples=spec["num_samples"] + 3 * k)) for k in range(5)]
    blobs = [c.blob for c in clips]
    num_tracks = spec["num_tracks"]
    parents = _hierarchy(rng, num_tracks)
    with runtime.Context(0) as ctx:
        handles = np.array([ctx.register_clip(b) for b in blobs], dtype=np.uint32)
        for handle in handles:
            ctx.set_clip_hierarchy(int(handle), parents)
        n = 40
        first = rng.integers(0, 5, size=n)
        others = rng.integers(0, 5, size=(n, num_blend - 1))
        times = np.array([rng.uniform(-0.05, clips[c].duration + 0.05) for c in first], dtype=np.float32)
        other_times = np.array([[rng.uniform(-0.05, clips[c].duration + 0.05) for c in row] for row in others], dtype=np.float32)
        weights = rng.dirichlet(np.ones(num_blend), size=n).astype(np.float32)
        weights[0] = 0.0
        weights[0, 0] = 1.0                                      # all the weight on the first clip
        weights[1] = 0.0
        weights[1, -1] = 1.0                                     # ... on the last
        base = rng.integers(0, 5, size=n)
        base_times = np.array([rng.uniform(0.0, clips[c].duration) for c in base], dtype=np.float32)
        base_poses = ob.oracle_decompress_tracks_batch(blobs, base, base_times, num_tracks)
        for rounding, looping, additive_format, object_space, base_as_buffer in (
                (0, 2, 0, False, False), (0, 2, 0, True, False), (3, 0, 1, True, False), (1, 1, 2, True, False), (2, 2, 3, False, False), (0, 2, 3, True, True), (0, 0, 1, False, True)):
            params = runtime.default_params(rounding_policy=rounding, looping_policy=looping)
            options = ob.default_options(looping_policy=looping)
            kwargs = dict(additive_format=additive_format, object_space=object_space, params=params,
                          blend_clips=handles[others], blend_sample_times=other_times, blend_weights=weights)
            if additive_format != 0 and base_as_buffer:
                buffer_poses = np.stack([ob.oracle_decompress_tracks(blobs[c], float(t), rounding, options) for c, t in zip(base, base_times)])
                kwargs.update(base_poses=buffer_poses)
            elif additive_format != 0:
                kwargs.update(base_clips=handles[base], base_sample_times=base_times)
            got = ctx.decompress_poses(handles[first], times, **kwargs)
            expected = ob.oracle_decompress_blended_poses_batch(blobs, first, times, others, other_times, weights, num_tracks, additive_format=additive_format,
                                                                base_clip_indices=base, base_sample_times=base_times, parent_indices=parents if object_space else None,
                                                                rounding=rounding, options=options)
            assert helpers.exact(got, expected), (name, num_blend, rounding, looping, additive_format, object_space, base_as_buffer)
        del base_poses
        assert ctx.rejected_instance_count() == 0


def test_blend_refuses_clips_of_another_size_and_bad_arguments():
    import torch
    device = torch.device("cuda:0")
    a, b, c = synth.build_clip(seed=801, num_tracks=30, num_samples=20), synth.build_clip(seed=802, num_tracks=30, num_samples=25), synth.build_clip(seed=803, num_tracks=31, num_samples=20)
    with runtime.Context(0) as ctx:
        ha, hb, hc = (ctx.register_clip(x.blob) for x in (a, b, c))
        n = 8
        sentinel = -7.25
        d_clips = torch.full((n,), ha, dtype=torch.int32, device=device)
        d_times = torch.full((n,), 0.25, dtype=torch.float32, device=device)
        others = np.full((n, 1), hb, dtype=np.int32)
        others[2, 0], others[5, 0] = hc, 12345                              # another track count, an unknown handle
        d_others = torch.from_numpy(others).to(device)
        d_other_times = torch.full((n, 1), 0.4, dtype=torch.float32, device=device)
        d_weights = torch.full((n, 2), 0.5, dtype=torch.float32, device=device)
        d_poses = torch.full((n, 31 * 12), sentinel, dtype=torch.float32, device=device)
        consumers = runtime.PoseConsumers()
        consumers.num_blend_clips = 2
        consumers.blend_clips, consumers.blend_sample_times, consumers.blend_weights = d_others.data_ptr(), d_other_times.data_ptr(), d_weights.data_ptr()
        ctx.decompress_poses_batch(d_clips.data_ptr(), d_times.data_ptr(), n, d_poses.data_ptr(), 31 * 48, consumers)
        torch.cuda.synchronize(device)
        poses = d_poses.cpu().numpy()
        assert ctx.rejected_instance_count() == 2
        assert np.all(poses[[2, 5]] == sentinel)
        expected = ob.oracle_blend_poses([ob.oracle_decompress_tracks(a.blob, 0.25), ob.oracle_decompress_tracks(b.blob, 0.4)], [0.5, 0.5])
        for i in (0, 1, 3, 4, 6, 7):
            assert helpers.exact(poses[i, : 30 * 12].reshape(30, 12), expected)
        consumers.num_blend_clips = 5
        with pytest.raises(runtime.AclHipError):
            ctx.decompress_poses_batch(d_clips.data_ptr(), d_times.data_ptr(), n, d_poses.data_ptr(), 31 * 48, consumers)
        consumers.num_blend_clips = 2
        consumers.blend_weights = None
        with pytest.raises(runtime.AclHipError):
            ctx.decompress_poses_batch(d_clips.data_ptr(), d_times.data_ptr(), n, d_poses.data_ptr(), 31 * 48, consumers)


def test_blend_at_full_size_every_instance():
    """65 536 instances, each the blend of three of 16 distinct 100-bone clips, object space: every pose compared with the oracle."""
    import torch
    device = torch.device("cuda:0")
    clips = [synth.build_clip(seed=900 + k, num_tracks=100, num_samples=61 + 7 * k, sample_rate=30.0) for k in range(16)]
    blobs = [c.blob for c in clips]
    parents = synth.humanoid_hierarchy(100)
    rng = np.random.default_rng(77)
    n = 65536
    with runtime.Context(0) as ctx:
        handles = np.array([ctx.register_clip(b) for b in blobs], dtype=np.int32)
        for handle in handles:
            ctx.set_clip_hierarchy(int(handle), parents)
        first, others = rng.integers(0, 16, size=n), rng.integers(0, 16, size=(n, 2))
        times, other_times = rng.uniform(0.0, 2.0, size=n).astype(np.float32), rng.uniform(0.0, 2.0, size=(n, 2)).astype(np.float32)
        weights = rng.dirichlet(np.ones(3), size=n).astype(np.float32)
        d = lambda array: torch.from_numpy(np.ascontiguousarray(array)).to(device)
        d_clips, d_times, d_others, d_other_times, d_weights = d(handles[first]), d(times), d(handles[others]), d(other_times), d(weights)
        d_poses = torch.zeros((n, 100, 12), dtype=torch.float32, device=device)
        consumers = runtime.PoseConsumers()
        consumers.object_space = 1
        consumers.num_blend_clips = 3
        consumers.blend_clips, consumers.blend_sample_times, consumers.blend_weights = d_others.data_ptr(), d_other_times.data_ptr(), d_weights.data_ptr()
        ctx.decompress_poses_batch(d_clips.data_ptr(), d_times.data_ptr(), n, d_poses.data_ptr(), 4800, consumers)
        torch.cuda.synchronize(device)
        expected = ob.oracle_decompress_blended_poses_batch(blobs, first, times, others, other_times, weights, 100, parent_indices=parents)
        assert helpers.bit_equal(d_poses.cpu().numpy(), expected)
        assert ctx.rejected_instance_count() == 0
