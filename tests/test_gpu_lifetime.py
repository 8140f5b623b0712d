"""Stream ordered clip lifetime: registering, replacing and unregistering clips while decodes are in flight never stalls or corrupts
them. The reference's contexts bind and reset in nanoseconds on the CPU (decompress.impl.h:66-83); an engine streams clips in and
out every frame, so the GPU side must not pay a device-wide synchronization for it. Needs a GPU."""
import os
import re
import threading
import time

import numpy as np
import pytest
import torch

from acl_amd import runtime, synth
from oracle import bindings as ob
import helpers

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_no_device_wide_synchronization_in_the_lifetime_paths():
    """source check: hipDeviceSynchronize / hipFree / default stream copies only where a call is synchronous by contract"""
    allowed = {"host_context.inl": ["aclhip_destroy"], "host_scalar_misc.inl": ["aclhip_get_rejected_instance_count", "aclhip_get_negative_scale_count"]}
    for name in ("host_clips.inl", "host_context.inl", "host_databases.inl", "host_consumers.inl", "host_launch.inl", "host_scalar_misc.inl"):
        text = open(os.path.join(ROOT, "acl_amd", "csrc", name)).read()
        count = len(re.findall(r"hipDeviceSynchronize\(", text))
        assert count == len(allowed.get(name, [])), (name, count)
        if name in ("host_clips.inl", "host_context.inl", "host_databases.inl", "host_launch.inl"):      # (the *_host convenience calls are synchronous by contract)
            assert not re.findall(r"\bhipMemcpy\(", text), name


def test_register_and_unregister_10000_clips_while_another_thread_decodes():
    rng = np.random.default_rng(0)
    resident = synth.build_clip(seed=2, num_tracks=100, num_samples=301, sample_rate=30.0)
    churn = [synth.build_clip(seed=700 + i, num_tracks=int(rng.integers(3, 130)), num_samples=int(rng.integers(2, 120)), has_scale=int(i % 3 == 0)) for i in range(64)]
    with runtime.Context(0) as context:
        handle = context.register_clip(resident.blob)
        device = torch.device("cuda", 0)
        n = 8192
        times = rng.uniform(0.0, resident.duration, size=n).astype(np.float32)
        expected = ob.oracle_decompress_tracks_batch([resident.blob], np.zeros(n, dtype=np.uint32), times, 100)
        d_clips = torch.full((n,), handle, dtype=torch.int32, device=device)
        d_times = torch.from_numpy(times).to(device)
        stats_before = context.lifetime_stats()

        stop = threading.Event()
        failures = []
        decoded_batches = [0]

        def decode_loop():
            stream = torch.cuda.Stream(device)
            with torch.cuda.stream(stream):
                d_poses = torch.zeros((n, 100, 12), dtype=torch.float32, device=device)
            stream.wait_stream(torch.cuda.current_stream(device))       # d_clips / d_times were uploaded on the current stream
            while not stop.is_set():
                for _ in range(8):
                    context.decompress_tracks_batch(d_clips.data_ptr(), d_times.data_ptr(), n, d_poses.data_ptr(), 4800, stream=stream.cuda_stream)
                stream.synchronize()
                poses = d_poses.cpu().numpy()
                if not np.array_equal(poses.view(np.uint32), expected.view(np.uint32)):
                    failures.append("pose mismatch while clips were registered / unregistered")
                    return
                decoded_batches[0] += 8
                with torch.cuda.stream(stream):
                    d_poses.zero_()

        worker = threading.Thread(target=decode_loop)
        worker.start()
        t0 = time.perf_counter()
        total = 10000
        live = []
        for i in range(total):
            live.append(context.register_clip(churn[i % len(churn)].blob, check_hash=False))
            if len(live) >= 48:
                # unregister in a scrambled order: slab pieces are freed out of order
                victims = sorted(rng.choice(len(live), size=32, replace=False), reverse=True)
                for index in victims:
                    context.unregister_clip(live.pop(int(index)))
        for clip in live:
            context.unregister_clip(clip)
        elapsed = time.perf_counter() - t0
        stop.set()
        worker.join()
        assert not failures, failures
        assert decoded_batches[0] > 0

        torch.cuda.synchronize(device)
        stats = context.lifetime_stats()
        assert stats["registered"] - stats_before["registered"] == total
        assert stats["unregistered"] - stats_before["unregistered"] == total
        assert stats["pending"] == 0 and stats["recycled"] >= total         # everything retired has been recycled (no leak)
        assert stats["table_address"] == stats_before["table_address"]      # the clip table did not move
        print(f"\n{total} clips registered + unregistered in {elapsed:.2f} s = {total / elapsed:.0f} clips/s next to {decoded_batches[0]} decoded batches "
              f"(virtual clip table: {stats['table_is_virtual']})")

        # the resident clip still decodes, and a fresh registration reuses recycled handles
        poses = context.decompress_tracks(np.full(16, handle), times[:16])
        assert helpers.exact(poses, expected[:16])
        again = context.register_clip(churn[0].blob)
        assert again < 1000         # (handles are recycled: 10 000 registrations never held more than a few hundred slots at once -- 250 with eight test processes sharing the device, under 200 alone)
        context.unregister_clip(again)
        context.unregister_clip(handle)


def test_clip_table_grows_in_place_past_its_first_pages():
    """20000 live clips: more than the 16384 records first backed; the table keeps its address (captured hipGraphs stay valid)"""
    clip = synth.build_clip(seed=5, num_tracks=3, num_samples=2)
    with runtime.Context(0) as context:
        first = context.register_clip(clip.blob)
        address = context.lifetime_stats()["table_address"]
        handles = [context.register_clip(clip.blob, check_hash=False) for _ in range(20000)]
        stats = context.lifetime_stats()
        assert stats["table_address"] == address
        assert stats["table_capacity"] >= 20001
        times = np.array([0.0, clip.duration], dtype=np.float32)
        a = context.decompress_tracks(np.full(2, first), times)
        b = context.decompress_tracks(np.full(2, handles[-1]), times)
        assert helpers.exact(a, b)
        for handle in handles[::-1]:
            context.unregister_clip(handle)
        context.unregister_clip(first)


@pytest.mark.parametrize("drain_between_rounds", [False, True])
def test_replacing_a_hierarchy_under_launches_in_flight(drain_between_rounds):
    """(drain_between_rounds: the replaced hierarchy's image has been recycled by the time the same hierarchy is set again -- round 3
    looked the shared image up BEFORE recycling and could hand the clip another skeleton's walk schedule)"""
    clip = synth.build_clip(seed=2, num_tracks=100, num_samples=301, sample_rate=30.0)
    with runtime.Context(0) as context:
        handle = context.register_clip(clip.blob)
        device = torch.device("cuda", 0)
        n = 4096
        rng = np.random.default_rng(3)
        times = rng.uniform(0.0, clip.duration, size=n).astype(np.float32)
        d_clips = torch.full((n,), handle, dtype=torch.int32, device=device)
        d_times = torch.from_numpy(times).to(device)
        d_poses = torch.zeros((n, 100, 12), dtype=torch.float32, device=device)
        consumers = runtime.PoseConsumers()
        consumers.object_space = 1
        stream = torch.cuda.Stream(device)
        stream.wait_stream(torch.cuda.current_stream(device))
        hierarchies = [synth.humanoid_hierarchy(100), np.concatenate([[runtime.NO_PARENT], np.arange(99)]).astype(np.uint32)]       # a humanoid, a chain
        local = ob.oracle_decompress_tracks_batch([clip.blob], np.zeros(n, dtype=np.uint32), times, 100)
        for round_index in range(6):
            parents = hierarchies[round_index % 2]
            context.set_clip_hierarchy(handle, parents)          # replaces the previous one while the last round's launches may still run
            for _ in range(4):
                context.decompress_poses_batch(d_clips.data_ptr(), d_times.data_ptr(), n, d_poses.data_ptr(), 4800, consumers, stream=stream.cuda_stream)
            if drain_between_rounds:
                stream.synchronize()
                assert helpers.bit_equal(d_poses.cpu().numpy(), ob.oracle_decompress_poses_batch([clip.blob], np.zeros(n, dtype=np.uint32), times, 100, parent_indices=parents)), round_index
        stream.synchronize()
        poses = d_poses.cpu().numpy()
        assert helpers.bit_equal(poses, ob.oracle_decompress_poses_batch([clip.blob], np.zeros(n, dtype=np.uint32), times, 100, parent_indices=hierarchies[1]))
        assert context.rejected_instance_count() == 0
        context.unregister_clip(handle)


def test_decodes_enqueued_before_an_unregistration_still_decode_the_clip():
    """A kernel reads the clip table when it EXECUTES. aclhip_unregister_clip therefore clears the clip's record behind the launches
    already enqueued (round 2 cleared it at once: a decode that had not started yet skipped the clip's instances). A long chain of
    launches keeps the stream busy, the clip is unregistered while most of them have not run, every pose of every launch is checked."""
    rng = np.random.default_rng(12)
    clip = synth.build_clip(seed=801, num_tracks=100, num_samples=120)
    filler = synth.build_clip(seed=802, num_tracks=300, num_samples=60, has_scale=1)
    device = torch.device("cuda", 0)
    with runtime.Context(0) as context:
        handle = context.register_clip(clip.blob)
        filler_handle = context.register_clip(filler.blob)
        n, launches = 4096, 24
        times = rng.uniform(0.0, clip.duration, size=n).astype(np.float32)
        expected = ob.oracle_decompress_tracks_batch([clip.blob], np.zeros(n, dtype=np.uint32), times, 100)
        stream = torch.cuda.Stream(device)
        d_clips = torch.full((n,), handle, dtype=torch.int32, device=device)
        d_times = torch.from_numpy(times).to(device)
        d_filler_clips = torch.full((65536,), filler_handle, dtype=torch.int32, device=device)
        d_filler_times = torch.zeros((65536,), dtype=torch.float32, device=device)
        d_filler_poses = torch.empty((65536, 300, 12), dtype=torch.float32, device=device)
        d_poses = torch.full((launches, n, 300, 12), -7.0, dtype=torch.float32, device=device)      # (rows of the largest registered pose)
        torch.cuda.synchronize(device)
        for k in range(launches):
            # ~0.2 ms of other work in front of every decode of the clip: the stream is far behind the host by the end of the loop
            context.decompress_tracks_batch(d_filler_clips.data_ptr(), d_filler_times.data_ptr(), 65536, d_filler_poses.data_ptr(), 300 * 48, stream=stream.cuda_stream)
            context.decompress_tracks_batch(d_clips.data_ptr(), d_times.data_ptr(), n, d_poses[k].data_ptr(), 300 * 48, stream=stream.cuda_stream)
        context.unregister_clip(handle)                     # returns at once: nothing waits for the stream
        still_running = not stream.query()
        stream.synchronize()
        poses = d_poses.cpu().numpy()
        for k in range(launches):
            assert helpers.bit_equal(poses[k, :, :100], expected), k
        assert context.rejected_instance_count() == 0
        assert still_running, "the stream had drained before the unregistration: the test did not exercise the ordering"
        # the handle is refused by launches made once the clear has happened
        context.decompress_tracks_batch(d_clips.data_ptr(), d_times.data_ptr(), n, d_poses[0].data_ptr(), 300 * 48, stream=stream.cuda_stream)
        stream.synchronize()
        assert context.rejected_instance_count() == n
        context.forget_stream(stream.cuda_stream)
        assert context.lifetime_stats()["launch_streams"] == 0
        context.unregister_clip(filler_handle)


def test_several_threads_launch_on_their_own_streams_next_to_a_churning_registry():
    """Launches hold the registry lock SHARED (round 5): four threads, each with its own stream, buffers and kind of batch -- whole poses,
    single bone requests, a compact layout, an instance list -- decode side by side while a fifth registers and unregisters clips
    (exclusive). Every result is the oracle's; a clip unregistered while the others launch is never decoded after its memory is gone
    (the unregistration's events cannot slip between a launch's noting its stream and enqueueing its kernel)."""
    rng = np.random.default_rng(5)
    resident = [synth.build_clip(seed=900 + i, num_tracks=100, num_samples=int(rng.integers(20, 200))) for i in range(6)]
    churn = [synth.build_clip(seed=950 + i, num_tracks=int(rng.integers(3, 130)), num_samples=int(rng.integers(2, 60))) for i in range(24)]
    with runtime.Context(0) as context:
        device = torch.device("cuda", 0)
        handles = np.array([context.register_clip(c.blob) for c in resident], dtype=np.uint32)
        durations = np.array([c.duration for c in resident], dtype=np.float32)
        n = 4096
        stop = threading.Event()
        failures = []
        rounds = [0, 0, 0, 0]

        def worker(kind):
            try:
                local = np.random.default_rng(100 + kind)
                which = local.integers(0, len(resident), size=n)
                times = (local.uniform(0.0, 1.0, size=n).astype(np.float32) * durations[which]).astype(np.float32)
                full = ob.oracle_decompress_tracks_batch([c.blob for c in resident], which, times, 100)
                stream = torch.cuda.Stream(device)
                with torch.cuda.stream(stream):
                    d_clips = torch.from_numpy(handles[which].astype(np.int32)).to(device)
                    d_times = torch.from_numpy(times).to(device)
                    tracks = local.integers(0, 100, size=n)
                    d_tracks = torch.from_numpy(tracks.astype(np.int32)).to(device)
                    width = {0: 1200, 1: 12, 2: 800, 3: 1200}[kind]
                    d_out = torch.zeros((n, width), dtype=torch.float32, device=device)
                stream.synchronize()
                expected = {0: full.reshape(n, 1200), 1: full[np.arange(n), tracks], 2: runtime.relayout_pose(full, runtime.LAYOUT_QV32).reshape(n, 800), 3: full.reshape(n, 1200)}[kind]
                output = runtime.OutputDesc()
                output.layout = runtime.LAYOUT_QV32
                instance_list = None
                if kind == 3:
                    instance_list = context.instance_list_create(n)
                    context.instance_list_set_clips(instance_list, d_clips.data_ptr(), stream=stream.cuda_stream)
                while not stop.is_set():
                    for _ in range(6):
                        if kind == 0:
                            context.decompress_tracks_batch(d_clips.data_ptr(), d_times.data_ptr(), n, d_out.data_ptr(), 4800, stream=stream.cuda_stream)
                        elif kind == 1:
                            context.decompress_track_batch(d_clips.data_ptr(), d_times.data_ptr(), d_tracks.data_ptr(), n, d_out.data_ptr(), stream=stream.cuda_stream)
                        elif kind == 2:
                            context.decompress_tracks_batch_out(d_clips.data_ptr(), d_times.data_ptr(), n, d_out.data_ptr(), 3200, output, stream=stream.cuda_stream)
                        else:
                            context.decompress_tracks_list(instance_list, d_times.data_ptr(), d_out.data_ptr(), 4800, poses_in_instance_order=True, stream=stream.cuda_stream)
                    stream.synchronize()
                    got = d_out.cpu().numpy()
                    if not np.array_equal(got.view(np.uint32), expected.view(np.uint32)):
                        failures.append(f"thread {kind}: poses differ from the oracle")
                        return
                    rounds[kind] += 1
                    with torch.cuda.stream(stream):
                        d_out.zero_()
                if instance_list is not None:
                    context.instance_list_destroy(instance_list)
            except Exception as error:      # noqa: BLE001 -- reported by the main thread
                failures.append(f"thread {kind}: {error!r}")

        threads = [threading.Thread(target=worker, args=(kind,)) for kind in range(4)]
        for thread in threads:
            thread.start()
        live = []
        deadline = time.perf_counter() + 4.0
        registered = 0
        while time.perf_counter() < deadline and not failures:
            live.append(context.register_clip(churn[registered % len(churn)].blob, check_hash=False))
            registered += 1
            if len(live) >= 16:
                for index in sorted(rng.choice(len(live), size=12, replace=False), reverse=True):
                    context.unregister_clip(live.pop(int(index)))
        stop.set()
        for thread in threads:
            thread.join()
        for clip in live:
            context.unregister_clip(clip)
        assert not failures, failures
        assert all(count > 0 for count in rounds), rounds
        assert context.rejected_instance_count() == 0
        print(f"\n{registered} clips registered next to {rounds} verified rounds of 6 launches per thread")
