"""aclhip_all_gather_poses: the optional gather of pose shards over RCCL. One GPU is available to the tests, so this drives a
single rank communicator (the gather then degenerates to a copy) -- the N > 1 layout is covered on CPU by tests/test_sharding_gloo.py."""
import ctypes
import os

import numpy as np
import pytest

from acl_amd import runtime, synth

pytestmark = pytest.mark.gpu


def _load_rccl():
    for name in ("librccl.so.1", "librccl.so"):
        try:
            return ctypes.CDLL(name, mode=ctypes.RTLD_GLOBAL)
        except OSError:
            continue
    return None


def test_single_rank_all_gather_copies_the_shard():
    import torch
    rccl = _load_rccl()
    if rccl is None:
        pytest.skip("librccl.so.1 not found")

    class UniqueId(ctypes.Structure):
        _fields_ = [("internal", ctypes.c_char * 128)]

    rccl.ncclGetUniqueId.argtypes = [ctypes.POINTER(UniqueId)]
    rccl.ncclCommInitRank.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, UniqueId, ctypes.c_int]
    rccl.ncclCommDestroy.argtypes = [ctypes.c_void_p]
    torch.cuda.set_device(0)
    unique_id = UniqueId()
    assert rccl.ncclGetUniqueId(ctypes.byref(unique_id)) == 0
    comm = ctypes.c_void_p()
    assert rccl.ncclCommInitRank(ctypes.byref(comm), 1, unique_id, 0) == 0

    context = runtime.Context(0)
    clip = synth.build_clip(seed=2, num_tracks=100, num_samples=60)
    handle = context.register_clip(clip.blob)
    n = 512
    times = np.random.default_rng(1).uniform(0.0, clip.duration, size=n).astype(np.float32)
    d_clips = torch.full((n,), handle, dtype=torch.int32, device="cuda")
    d_times = torch.from_numpy(times).cuda()
    shard = torch.zeros((n, 100, 12), dtype=torch.float32, device="cuda")
    gathered = torch.zeros_like(shard)
    stream = torch.cuda.current_stream()
    context.decompress_tracks_batch(d_clips.data_ptr(), d_times.data_ptr(), n, shard.data_ptr(), 4800, stream=stream.cuda_stream)
    context.all_gather_poses(comm, shard.data_ptr(), gathered.data_ptr(), shard.numel() * 4, stream=stream.cuda_stream)
    stream.synchronize()
    assert torch.equal(shard, gathered) and float(gathered.abs().sum()) > 0.0

    with pytest.raises(runtime.AclHipError):
        context.all_gather_poses(None, shard.data_ptr(), gathered.data_ptr(), 16)
    assert rccl.ncclCommDestroy(comm) == 0
    context.close()


def test_rccl_is_found_and_answers_without_a_communicator():
    """what the first N > 1 run on a real node would otherwise be the first to find out: the RCCL the library would call -- the one of
    THIS process (PyTorch's bundled librccl when torch.distributed made the communicator) -- is there, exports ncclAllGather and
    ncclGetVersion, and the version is one torch.distributed itself reports"""
    import torch
    version, path, how = runtime.probe_rccl()
    assert version >= 20000, version                 # RCCL 2.x (22606 = 2.26.6 on ROCm 7)
    assert "rccl" in os.path.basename(path), path
    print(f"RCCL {version} from {path} ({how})")
    reported = torch.cuda.nccl.version() if hasattr(torch.cuda, "nccl") else None
    if reported:
        major, minor, patch = reported[:3]
        # one RCCL per process: when torch has loaded its own, the probe must have found THAT one (same version), not a second library
        if "torch" in path or how != "loaded by this library":
            assert version == major * 10000 + minor * 100 + patch, (version, reported, path)
