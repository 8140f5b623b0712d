"""Multi-GPU sharding of a batch of clip instances: one process per GPU, instances are independent units.

The decode needs no exchange step: rank r decodes instances [r*n/W, (r+1)*n/W) of the batch into its own pose shard with its
own aclhip context (clip blobs are tiny and replicated on every GPU). Only when a caller wants the full pose set on
every rank is there a collective: one all-gather of the pose shards (RCCL over xGMI with backend "nccl", gloo on CPU).
SURVEY.md section 8(e); nothing comparable exists in the reference, which is single threaded CPU code.
"""
import torch
import torch.distributed as dist


def shard_bounds(num_instances, rank, world_size):
    """Contiguous, balanced partition of [0, num_instances): returns (begin, end) of `rank`'s shard."""
    if world_size <= 0 or not (0 <= rank < world_size):
        raise ValueError("invalid rank / world size")
    begin = (num_instances * rank) // world_size
    end = (num_instances * (rank + 1)) // world_size
    return begin, end


def shard_sizes(num_instances, world_size):
    return [shard_bounds(num_instances, r, world_size)[1] - shard_bounds(num_instances, r, world_size)[0] for r in range(world_size)]


def all_gather_poses(local_poses, num_instances, group=None):
    """Gathers the pose shards of every rank into one [num_instances, ...] tensor on every rank.

    local_poses: this rank's [shard, num_tracks, 12] (or [shard, floats]) tensor. Shards may differ by one instance:
    they are padded to the largest shard for the collective and trimmed afterwards.
    """
    world_size = dist.get_world_size(group)
    rank = dist.get_rank(group)
    sizes = shard_sizes(num_instances, world_size)
    if local_poses.shape[0] != sizes[rank]:
        raise ValueError(f"rank {rank} holds {local_poses.shape[0]} poses, its shard has {sizes[rank]}")

    largest = max(sizes)
    padded = local_poses
    if local_poses.shape[0] != largest:
        padded = torch.zeros((largest,) + tuple(local_poses.shape[1:]), dtype=local_poses.dtype, device=local_poses.device)
        padded[: local_poses.shape[0]] = local_poses
    padded = padded.contiguous()

    gathered = torch.empty((world_size * largest,) + tuple(local_poses.shape[1:]), dtype=local_poses.dtype, device=local_poses.device)
    dist.all_gather_into_tensor(gathered, padded, group=group)

    if all(size == largest for size in sizes):
        return gathered
    pieces = [gathered[r * largest: r * largest + sizes[r]] for r in range(world_size)]
    return torch.cat(pieces, dim=0)


class PeerGather:
    """Gather of every rank's pose shard onto ONE GPU (rank `dst`) by peer writes (aclhip_peer_* in the C ABI): the destination
    exports its buffer once (a 72 byte handle: HIP IPC handle + offset, broadcast over torch.distributed), every rank maps it and `push` copies its
    shard device to device to `rank * shard_bytes` -- each remote shard over its own xGMI link, all of the destination's links busy
    at once, nothing sent twice. `push` is asynchronous on `stream`; the destination may read `gathered` after every rank's stream has
    drained and the ranks have met (the caller's barrier). SURVEY 8(e)."""

    def __init__(self, context, shard_bytes, rank, world_size, dst=0, device=None, group=None):
        self.context, self.shard_bytes, self.rank, self.world_size, self.dst = context, int(shard_bytes), rank, world_size, dst
        self.gathered = None
        self.peer_ptr = None
        handle = torch.zeros(72, dtype=torch.uint8)       # ACLHIP_PEER_HANDLE_BYTES
        if rank == dst:
            self.gathered = torch.empty(world_size * self.shard_bytes, dtype=torch.uint8, device=device)
            exported = context.peer_export_buffer(self.gathered.data_ptr())
            if len(exported) != 72:
                raise ValueError("a peer handle is ACLHIP_PEER_HANDLE_BYTES = 72 bytes")
            handle = torch.frombuffer(bytearray(exported), dtype=torch.uint8).clone()
        on_device = dist.get_backend(group) == "nccl"
        if on_device:
            handle = handle.to(device)
        dist.broadcast(handle, src=dst, group=group)
        if rank == dst:
            self.target_ptr = self.gathered.data_ptr()
        else:
            self.peer_ptr = context.peer_open_buffer(bytes(handle.cpu().tolist()))
            self.target_ptr = self.peer_ptr

    def push(self, shard_ptr, stream=None):
        self.context.push_poses_to_peer(self.target_ptr, self.rank * self.shard_bytes, shard_ptr, self.shard_bytes, stream)

    def close(self):
        if self.peer_ptr is not None:
            self.context.peer_close_buffer(self.peer_ptr)
            self.peer_ptr = None


def stream_database_everywhere(context, database, tier, num_chunks, stream_in=True, src=0, group=None, stream=None):
    """Streamed database tiers are replicated per GPU (every rank registers the database with its own context), so their residency
    must advance identically everywhere: rank `src` decides the request -- (tier, num_chunks, in or out), the arguments of the other
    ranks are ignored -- it is broadcast, every rank applies it to its own context (aclhip_database_stream_in / _out, asynchronous
    on `stream`), and the ranks check that they moved the same number of chunks. Returns that number.
    SURVEY.md section 8(e): "streaming state advanced identically on all ranks (broadcast of 'chunks [a, b) now resident')"."""
    device = torch.device("cuda", context.device_index) if dist.get_backend(group) == "nccl" else torch.device("cpu")
    request = torch.tensor([int(tier), int(num_chunks), 1 if stream_in else 0], dtype=torch.int64, device=device)
    dist.broadcast(request, src=src, group=group)
    tier, num_chunks, stream_in = (int(v) for v in request.tolist())
    apply = context.database_stream_in if stream_in else context.database_stream_out
    moved = int(apply(database, tier, num_chunks, stream=stream))

    spread = torch.tensor([moved, -moved], dtype=torch.int64, device=device)
    dist.all_reduce(spread, op=dist.ReduceOp.MAX, group=group)
    if int(spread[0]) != -int(spread[1]):
        raise RuntimeError(f"database residency diverged between ranks: this rank moved {moved} chunks, others between {-int(spread[1])} and {int(spread[0])}")
    return moved
