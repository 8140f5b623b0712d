"""Persistent instance lists (aclhip_instance_list_*): the library keeps an instance list in decode (locality) order across frames --
ordered once, patched in place when instances change clip, re-ordered before a decode once an eighth of it has changed -- and
decodes it with the frame's sample times gathered through that order. Every pose of every frame against the oracle. Needs a GPU."""
import ctypes
import os

import numpy as np
import pytest
import torch

from acl_amd import runtime, synth
from oracle import bindings as ob
import helpers
from test_order_instances import check_order

pytestmark = pytest.mark.gpu


_hip = None


def _read_order(address, n):
    """the list's slot -> instance order (a device address the library owns) copied to the host"""
    global _hip
    if _hip is None:
        _hip = ctypes.CDLL("libamdhip64.so")
        _hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
    out = np.zeros(n, dtype=np.uint32)
    assert _hip.hipMemcpy(out.ctypes.data, address, n * 4, 2) == 0      # hipMemcpyDeviceToHost
    return out


@pytest.mark.parametrize("num_instances,num_clips,big_rig_tracks", [(1, 1, 0), (5000, 7, 0), (65536, 256, 0), (6000, 9, 300)])
def test_list_decodes_follow_the_oracle_through_updates_and_reorders(num_instances, num_clips, big_rig_tracks):
    rng = np.random.default_rng(num_instances + num_clips)
    clips = [synth.build_clip(seed=500 + i, num_tracks=int(rng.integers(40, 101)), num_samples=int(rng.integers(2, 90)), has_scale=int(i % 5 == 0)) for i in range(num_clips)]
    if big_rig_tracks:
        clips[0] = synth.build_clip(seed=499, num_tracks=big_rig_tracks, num_samples=40, has_scale=1)
    blobs = [c.blob for c in clips]
    durations = np.array([c.duration for c in clips], dtype=np.float32)
    max_tracks = max(c.num_tracks for c in clips)
    windows = -(-max_tracks * 3 // 312)
    n = num_instances
    device = torch.device("cuda", 0)
    with runtime.Context(0) as context:
        handles = np.array([context.register_clip(c.blob) for c in clips], dtype=np.uint32)
        stream = torch.cuda.Stream(device)
        which = rng.integers(0, num_clips, size=n)
        d_clips = torch.from_numpy(handles[which].astype(np.int32)).to(device)
        d_times = torch.zeros(n, dtype=torch.float32, device=device)
        d_poses = torch.zeros((n, max_tracks, 12), dtype=torch.float32, device=device)
        d_rows = torch.zeros((n, max_tracks, 12), dtype=torch.float32, device=device)
        torch.cuda.synchronize(device)

        instance_list = context.instance_list_create(n)
        with pytest.raises(runtime.AclHipError):       # nothing to decode before the clips are set
            context.decompress_tracks_list(instance_list, d_times.data_ptr(), d_poses.data_ptr(), max_tracks * 48, stream=stream.cuda_stream)
        context.instance_list_set_clips(instance_list, d_clips.data_ptr(), stream=stream.cuda_stream)
        order_address, orderings = context.instance_list_order(instance_list)
        assert orderings == 1

        changed_total = 0
        for frame in range(12):
            # a few instances change clip (1 % per frame; more in one frame so that a re-order happens inside the test's frames)
            num_changes = max(1, n // 100) if frame != 6 else max(1, n // 20)
            changed = rng.choice(n, size=min(num_changes, n), replace=False).astype(np.int32)
            new_which = rng.integers(0, num_clips, size=changed.size)
            which[changed] = new_which
            d_changed = torch.from_numpy(changed).to(device)
            d_new = torch.from_numpy(handles[new_which].astype(np.int32)).to(device)
            torch.cuda.synchronize(device)
            context.instance_list_update(instance_list, d_changed.data_ptr(), d_new.data_ptr(), changed.size, stream=stream.cuda_stream)
            changed_total += changed.size

            times = (rng.uniform(0.0, 1.0, size=n).astype(np.float32) * durations[which]).astype(np.float32)
            with torch.cuda.stream(stream):
                d_times.copy_(torch.from_numpy(times), non_blocking=False)
                d_poses.zero_()         # (rows past a clip's own tracks keep what the buffer held: the oracle's rows are zero there)
                d_rows.zero_()
            context.decompress_tracks_list(instance_list, d_times.data_ptr(), d_poses.data_ptr(), max_tracks * 48, stream=stream.cuda_stream)
            context.decompress_tracks_list(instance_list, d_times.data_ptr(), d_rows.data_ptr(), max_tracks * 48, poses_in_instance_order=True, stream=stream.cuda_stream)
            stream.synchronize()

            order = _read_order(order_address, n)
            assert np.array_equal(np.sort(order), np.arange(n, dtype=np.uint32))
            expected = ob.oracle_decompress_tracks_batch(blobs, which.astype(np.uint32), times, max_tracks)
            assert helpers.bit_equal(d_rows.cpu().numpy(), expected), frame
            assert helpers.bit_equal(d_poses.cpu().numpy(), expected[order]), frame
        _, orderings = context.instance_list_order(instance_list)
        assert orderings >= 2 or n < 16, orderings          # the list was re-ordered along the way ...
        assert orderings <= 6 or n < 100, orderings         # ... but not every frame
        # right after a re-order the list is in locality order for the clips the instances play NOW
        context.instance_list_set_clips(instance_list, torch.from_numpy(handles[which].astype(np.int32)).to(device).data_ptr(), stream=stream.cuda_stream)
        stream.synchronize()
        check_order(handles[which], _read_order(order_address, n), windows, stable=False)
        assert context.rejected_instance_count() == 0
        context.instance_list_destroy(instance_list)
        with pytest.raises(runtime.AclHipError):
            context.decompress_tracks_list(instance_list, d_times.data_ptr(), d_poses.data_ptr(), max_tracks * 48, stream=stream.cuda_stream)


def test_single_launch_order_equals_the_three_launch_order_in_structure():
    """aclhip_order_instances_device at three batch sizes -- 1, 64 and 64 (larger) workgroups of the one launch form --, the same scratch
    call after call: every result is a valid locality order."""
    rng = np.random.default_rng(3)
    clips = [synth.build_clip(seed=600 + i, num_tracks=50, num_samples=20) for i in range(40)]
    with runtime.Context(0) as context:
        handles = np.array([context.register_clip(c.blob) for c in clips], dtype=np.uint32)
        device = torch.device("cuda", 0)
        for n in (100, 70000, 600000):
            which = rng.integers(0, len(clips), size=n)
            d_clips = torch.from_numpy(handles[which].astype(np.int32)).to(device)
            d_order = torch.zeros(n, dtype=torch.int32, device=device)
            for _ in range(3):                  # the same scratch, call after call
                context.order_instances_device(d_clips.data_ptr(), 0, n, d_order.data_ptr(), 0, 0)
                torch.cuda.synchronize(device)
                check_order(handles[which], d_order.cpu().numpy().astype(np.uint32), 1, stable=False)


@pytest.mark.parametrize("layout", ["qvv48", "qvv40", "qv32"])
def test_list_decodes_take_output_descriptors(layout):
    """aclhip_decompress_tracks_list with an aclhip_output_desc -- pose layout, a skipped sub-track kind, per track skips --, in slot
    order and in instance order: the same bytes as aclhip_decompress_tracks_batch_out writes for the same instances"""
    rng = np.random.default_rng(7)
    clips = [synth.build_clip(seed=700 + i, num_tracks=60, num_samples=30, has_scale=1) for i in range(9)]
    layout_id, bytes_per_track = runtime.LAYOUTS[layout]
    n, tracks = 3000, 60
    floats_per_track = bytes_per_track // 4
    device = torch.device("cuda", 0)
    with runtime.Context(0) as context:
        handles = np.array([context.register_clip(c.blob) for c in clips], dtype=np.uint32)
        which = rng.integers(0, len(clips), size=n)
        times = np.array([rng.uniform(0.0, clips[w].duration) for w in which], dtype=np.float32)
        d_clips = torch.from_numpy(handles[which].astype(np.int32)).to(device)
        d_times = torch.from_numpy(times).to(device)
        skips = rng.integers(0, 8, size=tracks).astype(np.uint8)          # bit 0 / 1 / 2: rotation / translation / scale of that track
        d_skips = torch.from_numpy(skips).to(device)
        output = runtime.OutputDesc()
        output.layout = layout_id
        output.skip_translations = 1
        output.skip_tracks = d_skips.data_ptr()
        fill = 123.25
        d_direct = torch.full((n, tracks, floats_per_track), fill, dtype=torch.float32, device=device)
        d_slots = torch.full((n, tracks, floats_per_track), fill, dtype=torch.float32, device=device)
        d_rows = torch.full((n, tracks, floats_per_track), fill, dtype=torch.float32, device=device)
        torch.cuda.synchronize(device)
        context.decompress_tracks_batch_out(d_clips.data_ptr(), d_times.data_ptr(), n, d_direct.data_ptr(), tracks * bytes_per_track, output)
        instance_list = context.instance_list_create(n)
        context.instance_list_set_clips(instance_list, d_clips.data_ptr())
        context.decompress_tracks_list(instance_list, d_times.data_ptr(), d_slots.data_ptr(), tracks * bytes_per_track, output=output)
        context.decompress_tracks_list(instance_list, d_times.data_ptr(), d_rows.data_ptr(), tracks * bytes_per_track, output=output, poses_in_instance_order=True)
        torch.cuda.synchronize(device)
        order = _read_order(context.instance_list_order(instance_list)[0], n)
        direct = d_direct.cpu().numpy()
        assert np.any(direct != fill) and np.any(direct == fill)         # something was written, something was skipped
        assert np.array_equal(d_rows.cpu().numpy().view(np.uint32), direct.view(np.uint32))
        assert np.array_equal(d_slots.cpu().numpy().view(np.uint32), direct[order].view(np.uint32))
        with pytest.raises(runtime.AclHipError):                        # the list decides the rows itself
            output.rows = d_clips.data_ptr()
            context.decompress_tracks_list(instance_list, d_times.data_ptr(), d_slots.data_ptr(), tracks * bytes_per_track, output=output)
        context.instance_list_destroy(instance_list)


def test_a_captured_list_decode_replays_with_new_sample_times():
    """aclhip_decompress_tracks_list captured into a hipGraph: the replays decode the list in the order it had when it was captured,
    with whatever sample times the buffer holds by then (the header says so: when to re-order is decided on the host, at call time)"""
    rng = np.random.default_rng(11)
    clips = [synth.build_clip(seed=800 + i, num_tracks=80, num_samples=40) for i in range(12)]
    blobs = [c.blob for c in clips]
    durations = np.array([c.duration for c in clips], dtype=np.float32)
    n, tracks = 8192, 80
    device = torch.device("cuda", 0)
    with runtime.Context(0) as context:
        handles = np.array([context.register_clip(c.blob) for c in clips], dtype=np.uint32)
        which = rng.integers(0, len(clips), size=n)
        stream = torch.cuda.Stream(device)
        d_clips = torch.from_numpy(handles[which].astype(np.int32)).to(device)
        d_times = torch.zeros(n, dtype=torch.float32, device=device)
        d_rows = torch.zeros((n, tracks, 12), dtype=torch.float32, device=device)
        torch.cuda.synchronize(device)
        instance_list = context.instance_list_create(n)
        context.instance_list_set_clips(instance_list, d_clips.data_ptr(), stream=stream.cuda_stream)
        context.decompress_tracks_list(instance_list, d_times.data_ptr(), d_rows.data_ptr(), tracks * 48, poses_in_instance_order=True, stream=stream.cuda_stream)
        stream.synchronize()

        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=stream):
            context.decompress_tracks_list(instance_list, d_times.data_ptr(), d_rows.data_ptr(), tracks * 48, poses_in_instance_order=True, stream=stream.cuda_stream)
        for replay in range(4):
            times = (rng.uniform(0.0, 1.0, size=n).astype(np.float32) * durations[which]).astype(np.float32)
            d_times.copy_(torch.from_numpy(times))
            torch.cuda.synchronize(device)
            graph.replay()
            torch.cuda.synchronize(device)
            expected = ob.oracle_decompress_tracks_batch(blobs, which.astype(np.uint32), times, tracks)
            assert helpers.bit_equal(d_rows.cpu().numpy(), expected), replay
        del graph
        context.instance_list_destroy(instance_list)


def test_an_attached_list_decodes_the_callers_own_clip_array():
    """aclhip_instance_list_attach: no copy of the clips, no update launch -- the caller writes clip changes into its own device array,
    tells the list how many (aclhip_instance_list_note_changes) and every decode follows the array as it is THEN, in the list's order
    or in the caller's rows; the list re-orders itself from the array once an eighth has changed"""
    rng = np.random.default_rng(21)
    with runtime.Context(0) as context:
        clips = [synth.build_clip(seed=800 + i, num_tracks=int(rng.integers(20, 101)), num_samples=int(rng.integers(2, 90))) for i in range(23)]
        handles = np.array([context.register_clip(c.blob) for c in clips], dtype=np.uint32)
        durations = np.array([c.duration for c in clips], dtype=np.float32)
        max_tracks = 100
        n = 9000
        device = torch.device("cuda", 0)
        which = rng.integers(0, len(clips), size=n)
        d_clips = torch.from_numpy(handles[which].astype(np.int32)).to(device)          # the CALLER's array: the list keeps reading it
        instance_list = context.instance_list_create(n)
        context.instance_list_attach(instance_list, d_clips.data_ptr())
        order_address, orderings = context.instance_list_order(instance_list)
        assert orderings == 1
        with pytest.raises(runtime.AclHipError):                                        # an attached list has no update call
            context.instance_list_update(instance_list, d_clips.data_ptr(), d_clips.data_ptr(), 1)
        for frame in range(10):
            count = n // 100 if frame != 5 else n // 5
            changed = rng.choice(n, size=count, replace=False)
            which[changed] = rng.integers(0, len(clips), size=count)
            d_clips.index_copy_(0, torch.from_numpy(changed).to(device), torch.from_numpy(handles[which[changed]].astype(np.int32)).to(device))      # the caller's kernel
            context.instance_list_note_changes(instance_list, count)
            times = (rng.uniform(0.0, 1.0, size=n).astype(np.float32) * durations[which]).astype(np.float32)
            d_times = torch.from_numpy(times).to(device)
            expected = ob.oracle_decompress_tracks_batch([c.blob for c in clips], which, times, max_tracks)
            d_rows = torch.zeros((n, max_tracks, 12), dtype=torch.float32, device=device)
            d_slots = torch.zeros((n, max_tracks, 12), dtype=torch.float32, device=device)
            context.decompress_tracks_list(instance_list, d_times.data_ptr(), d_rows.data_ptr(), max_tracks * 48, poses_in_instance_order=True)
            context.decompress_tracks_list(instance_list, d_times.data_ptr(), d_slots.data_ptr(), max_tracks * 48)
            torch.cuda.synchronize(device)
            order = _read_order(order_address, n)
            assert helpers.bit_equal(d_rows.cpu().numpy(), expected), frame
            assert helpers.bit_equal(d_slots.cpu().numpy(), expected[order]), frame
        _, orderings = context.instance_list_order(instance_list)
        assert 2 <= orderings <= 4, orderings          # re-ordered when a fifth changed at once, not every frame
        # back to a list of its own: set_clips detaches
        context.instance_list_set_clips(instance_list, d_clips.data_ptr())
        context.instance_list_update(instance_list, torch.zeros(1, dtype=torch.int32, device=device).data_ptr(), d_clips.data_ptr(), 1)
        with pytest.raises(runtime.AclHipError):
            context.instance_list_note_changes(instance_list, 1)
        assert context.rejected_instance_count() == 0
        context.instance_list_destroy(instance_list)
