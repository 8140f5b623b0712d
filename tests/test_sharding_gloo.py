"""N > 1 path on CPU: two processes over gloo shard a batch of instances, decode their shard and all-gather the poses.
The decode stand-in on CPU is the oracle (test infrastructure) -- the point here is the sharding / gather logic."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from acl_amd import sharding


def test_shard_bounds_partition_everything():
    for n in (0, 1, 7, 64, 65536, 65537):
        for world in (1, 2, 3, 8):
            bounds = [sharding.shard_bounds(n, r, world) for r in range(world)]
            assert bounds[0][0] == 0 and bounds[-1][1] == n
            assert all(bounds[r][1] == bounds[r + 1][0] for r in range(world - 1))
            sizes = [e - b for b, e in bounds]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        sharding.shard_bounds(10, 2, 2)


def _worker(rank, world_size, port, num_instances, result_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world_size)
    try:
        import sys
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        from acl_amd import synth
        from oracle import bindings as ob

        clip = synth.build_clip(seed=9, num_tracks=13, num_samples=50)
        rng = np.random.default_rng(99)                  # same instance list on every rank
        times = rng.uniform(0.0, clip.duration, size=num_instances).astype(np.float32)

        begin, end = sharding.shard_bounds(num_instances, rank, world_size)
        local = np.zeros((end - begin, clip.num_tracks, 12), dtype=np.float32)
        for i in range(begin, end):
            local[i - begin] = ob.oracle_decompress_tracks(clip.blob, float(times[i]))

        gathered = sharding.all_gather_poses(torch.from_numpy(local), num_instances)
        assert gathered.shape == (num_instances, clip.num_tracks, 12)

        full = np.stack([ob.oracle_decompress_tracks(clip.blob, float(t)) for t in times]) if num_instances else np.zeros((0, clip.num_tracks, 12), np.float32)
        assert np.array_equal(gathered.numpy(), full)
        open(os.path.join(result_dir, f"ok{rank}"), "w").close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("num_instances", [10, 11])
def test_two_ranks_shard_and_all_gather(tmp_path, num_instances):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_worker, args=(2, port, num_instances, str(tmp_path)), nprocs=2, join=True)
    assert os.path.exists(tmp_path / "ok0") and os.path.exists(tmp_path / "ok1")


def test_eight_ranks_shard_and_all_gather(tmp_path):
    """the node shape of BASELINE.json configs[3] / [4]: 8 ranks, shards that differ by one instance (43 = 8 * 5 + 3)"""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_worker, args=(8, port, 43, str(tmp_path)), nprocs=8, join=True)
    assert all(os.path.exists(tmp_path / f"ok{rank}") for rank in range(8))


class _PeerContext:
    """stands in for runtime.Context on a box without GPUs: "device buffers" are host tensors, a peer handle is the name of a file
    every process can map (what the HIP IPC handle is on the GPU box)"""

    def __init__(self, directory):
        self.directory = directory
        self.mapped = {}

    def peer_export_buffer(self, buffer_ptr):
        name = b"gathered.bin"
        return name + b"\0" * (72 - len(name))

    def peer_open_buffer(self, handle):
        path = os.path.join(self.directory, handle.rstrip(b"\0").decode())
        mapped = np.memmap(path, dtype=np.uint8, mode="r+")
        self.mapped[id(mapped)] = mapped
        return id(mapped)

    def peer_close_buffer(self, pointer):
        self.mapped.pop(pointer).flush()

    def push_poses_to_peer(self, peer_ptr, offset_bytes, shard, shard_bytes, stream=None):
        self.mapped[peer_ptr][offset_bytes: offset_bytes + shard_bytes] = shard
        self.mapped[peer_ptr].flush()


def _peer_worker(rank, world_size, port, result_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world_size)
    try:
        shard_bytes = 480
        path = os.path.join(result_dir, "gathered.bin")
        context = _PeerContext(result_dir)
        if rank == 0:
            np.zeros(world_size * shard_bytes, dtype=np.uint8).tofile(path)
        dist.barrier()
        gather = sharding.PeerGather(context, shard_bytes, rank, world_size, dst=0, device="cpu")
        if rank == 0:
            # the destination writes through its own buffer: stand in for it with the same mapping the others use
            gather.target_ptr = context.peer_open_buffer(context.peer_export_buffer(0))
        shard = np.full(shard_bytes, rank + 1, dtype=np.uint8)
        gather.push(shard)
        dist.barrier()
        if rank == 0:
            gathered = np.fromfile(path, dtype=np.uint8).reshape(world_size, shard_bytes)
            assert (gathered == np.arange(1, world_size + 1, dtype=np.uint8)[:, None]).all()
        gather.close()
        open(os.path.join(result_dir, f"peer_ok{rank}"), "w").close()
    finally:
        dist.destroy_process_group()


def test_peer_gather_protocol_places_every_shard(tmp_path):
    """sharding.PeerGather: handle broadcast from the destination, every rank pushes its shard to rank * shard_bytes"""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_peer_worker, args=(4, port, str(tmp_path)), nprocs=4, join=True)
    assert all(os.path.exists(tmp_path / f"peer_ok{rank}") for rank in range(4))


class _RecordingContext:
    """stands in for runtime.Context on a box without GPUs: a database of 5 + 3 chunks whose residency is tracked like
    aclhip_database_stream_in / _out would (first `loaded` chunks of a tier are resident)"""
    device_index = 0

    def __init__(self, skew=0):
        self.loaded = {1: 0, 2: 0}
        self.total = {1: 5, 2: 3}
        self.calls = []
        self.skew = skew

    def database_stream_in(self, database, tier, num_chunks, stream=None):
        moved = max(min(num_chunks, self.total[tier] - self.loaded[tier]) - self.skew, 0)
        self.loaded[tier] += moved
        self.calls.append(("in", tier, num_chunks))
        return moved

    def database_stream_out(self, database, tier, num_chunks, stream=None):
        moved = min(num_chunks, self.loaded[tier])
        self.loaded[tier] -= moved
        self.calls.append(("out", tier, num_chunks))
        return moved


def _streaming_worker(rank, world_size, port, result_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group(backend="gloo", rank=rank, world_size=world_size)
    try:
        context = _RecordingContext()
        # only rank 0's arguments count: rank 1 passes nonsense and still ends up in the same state
        script = [(1, 2, True), (2, 0xFFFFFFFF, True), (1, 1, False), (1, 0xFFFFFFFF, True)]
        moved = []
        for tier, num_chunks, stream_in in script:
            mine = (tier, num_chunks, stream_in) if rank == 0 else (2, 12345, not stream_in)
            moved.append(sharding.stream_database_everywhere(context, 0, *mine, src=0))
        assert moved == [2, 3, 1, 4]
        assert context.loaded == {1: 5, 2: 3}
        assert context.calls == [("in", 1, 2), ("in", 2, 0xFFFFFFFF), ("out", 1, 1), ("in", 1, 0xFFFFFFFF)]

        # a rank whose residency diverges is noticed by everyone
        diverging = _RecordingContext(skew=1 if rank == 1 else 0)
        try:
            sharding.stream_database_everywhere(diverging, 0, 1, 3, True, src=0)
            raised = False
        except RuntimeError:
            raised = True
        assert raised
        open(os.path.join(result_dir, f"stream_ok{rank}"), "w").close()
    finally:
        dist.destroy_process_group()


def test_two_ranks_advance_database_residency_together(tmp_path):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_streaming_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert os.path.exists(tmp_path / "stream_ok0") and os.path.exists(tmp_path / "stream_ok1")
