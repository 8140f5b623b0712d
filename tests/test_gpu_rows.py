"""aclhip_decompress_tracks_batch_rows: the decode order of a batch is independent of where its poses go. Needs a GPU."""
import numpy as np
import pytest
import torch

from acl_amd import runtime, synth
from oracle import bindings as ob
import helpers

pytestmark = pytest.mark.gpu


def launch(context, handles, times, rows, num_rows, num_tracks, params=None):
    d_handles = torch.from_numpy(handles.astype(np.int32)).cuda()
    d_times = torch.from_numpy(times).cuda()
    poses = torch.full((num_rows, num_tracks, 12), 7.0, dtype=torch.float32, device="cuda")
    if rows is None:
        context.decompress_tracks_batch(d_handles.data_ptr(), d_times.data_ptr(), handles.size, poses.data_ptr(), num_tracks * 48, params=params)
    else:
        d_rows = torch.from_numpy(rows.astype(np.int32)).cuda()
        context.decompress_tracks_batch_rows(d_handles.data_ptr(), d_times.data_ptr(), d_rows.data_ptr(), handles.size, poses.data_ptr(), num_tracks * 48, params=params)
    torch.cuda.synchronize()
    return poses.cpu().numpy()


@pytest.mark.parametrize("num_tracks", [40, 100, 250])
def test_locality_order_leaves_the_poses_where_they_were(num_tracks):
    """many clips, decoded in aclhip_order_instances_for_locality order with rows = the order: the pose buffer is bit identical
    to the plain launch (one and several wavefronts per pose)"""
    rng = np.random.default_rng(num_tracks)
    with runtime.Context(0) as context:
        clips = [synth.build_clip(seed=500 + i, num_tracks=num_tracks, num_samples=int(rng.integers(20, 90)), has_scale=i % 2) for i in range(24)]
        handles = np.array([context.register_clip(c.blob) for c in clips], dtype=np.uint32)
        n = 3001
        which = rng.integers(0, len(clips), size=n)
        times = np.array([rng.uniform(0.0, clips[c].duration) for c in which], dtype=np.float32)
        plain = launch(context, handles[which], times, None, n, num_tracks)

        order = context.order_instances_for_locality(handles[which])
        assert np.array_equal(np.sort(order), np.arange(n))
        ordered = launch(context, handles[which][order], times[order], order, n, num_tracks)
        assert helpers.exact(ordered, plain)
        for i in rng.integers(0, n, size=16):
            assert helpers.exact(plain[i], ob.oracle_decompress_tracks(clips[which[i]].blob, float(times[i])))

        # the any-settings kernel takes rows too
        params = runtime.default_params(normalization=runtime.NORMALIZE_ALWAYS)
        assert helpers.exact(launch(context, handles[which][order], times[order], order, n, num_tracks, params), launch(context, handles[which], times, None, n, num_tracks, params))
        assert context.rejected_instance_count() == 0


def test_rows_scatter_into_a_larger_buffer():
    """rows need not be a permutation of the instance indices: any distinct rows of the caller's buffer; the rest stays untouched"""
    rng = np.random.default_rng(4)
    with runtime.Context(0) as context:
        clip = synth.build_clip(seed=71, num_tracks=33, num_samples=50)
        handle = context.register_clip(clip.blob)
        n, num_rows = 100, 257
        times = rng.uniform(0.0, clip.duration, size=n).astype(np.float32)
        rows = rng.choice(num_rows, size=n, replace=False).astype(np.uint32)
        poses = launch(context, np.full(n, handle, dtype=np.uint32), times, rows, num_rows, 33)
        for i in range(n):
            assert helpers.exact(poses[rows[i]], ob.oracle_decompress_tracks(clip.blob, float(times[i])))
        untouched = np.setdiff1d(np.arange(num_rows), rows)
        assert np.all(poses[untouched] == 7.0)


def test_clip_memory_is_recycled_across_registrations():
    """clips share 32 MB slabs (bump allocation, a slab is recycled when its last clip goes): churn must neither leak slabs nor
    disturb clips that stay"""
    rng = np.random.default_rng(8)
    with runtime.Context(0) as context:
        keeper = synth.build_clip(seed=900, num_tracks=30, num_samples=40)
        keeper_handle = context.register_clip(keeper.blob)
        keeper_times = rng.uniform(0.0, keeper.duration, size=8).astype(np.float32)
        expected = context.decompress_tracks(np.full(8, keeper_handle, dtype=np.uint32), keeper_times)
        free_before = torch.cuda.mem_get_info()[0]
        big = synth.build_clip(seed=901, num_tracks=2500, num_samples=400)       # larger than half a slab? no: exercises a large piece
        for generation in range(6):
            clips = [synth.build_clip(seed=1000 + generation * 40 + i, num_tracks=60, num_samples=200) for i in range(40)] + [big]
            handles = [context.register_clip(c.blob) for c in clips]
            probe = int(rng.integers(0, len(clips)))
            t = np.array([clips[probe].duration * 0.37], dtype=np.float32)
            got = context.decompress_tracks(np.array([handles[probe]], dtype=np.uint32), t)
            assert helpers.exact(got[0], ob.oracle_decompress_tracks(clips[probe].blob, float(t[0])))
            for handle in handles:
                context.unregister_clip(handle)
            assert helpers.exact(context.decompress_tracks(np.full(8, keeper_handle, dtype=np.uint32), keeper_times), expected)
        # six generations of 41 clips did not grow the footprint by more than a couple of slabs
        assert free_before - torch.cuda.mem_get_info()[0] < 3 * 32 * 1024 * 1024
