"""The N > 1 path with real device memory on the ONE GPU the test box has: several processes share cuda:0 (gloo carries the control
messages), every rank decodes its shard with its own aclhip context and the shards are gathered
  * by peer writes into rank 0's buffer (aclhip_peer_* / sharding.PeerGather: HIP IPC mapping + device to device copies), and
  * through bench.py's own N > 1 control flow (ACLHIP_BENCH_BACKEND=gloo dry run of `torchrun bench.py --gpus N`).
On an 8-GPU node the same code runs one rank per GPU over RCCL / xGMI. Needs a GPU."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _peer_worker(rank, world_size, port, result_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world_size)
    try:
        from acl_amd import runtime, sharding, synth
        from oracle import bindings as ob

        clip = synth.build_clip(seed=9, num_tracks=61, num_samples=50, has_scale=1)
        num_instances = 1000 * world_size
        rng = np.random.default_rng(99)                  # same instance list on every rank
        times = rng.uniform(0.0, clip.duration, size=num_instances).astype(np.float32)
        begin, end = sharding.shard_bounds(num_instances, rank, world_size)

        context = runtime.Context(0)
        handle = context.register_clip(clip.blob)
        device = torch.device("cuda", 0)
        d_clips = torch.full((end - begin,), handle, dtype=torch.int32, device=device)
        d_times = torch.from_numpy(times[begin:end]).to(device)
        d_shard = torch.zeros((end - begin, clip.num_tracks, 12), dtype=torch.float32, device=device)
        stream = torch.cuda.current_stream(device)
        shard_bytes = d_shard.numel() * 4
        gather = sharding.PeerGather(context, shard_bytes, rank, world_size, dst=0, device=device)
        for _ in range(2):      # the mapping is reused batch after batch
            context.decompress_tracks_batch(d_clips.data_ptr(), d_times.data_ptr(), end - begin, d_shard.data_ptr(), clip.num_tracks * 48, stream=stream.cuda_stream)
            gather.push(d_shard.data_ptr(), stream=stream.cuda_stream)
            stream.synchronize()
            dist.barrier()
            if rank == 0:
                gathered = gather.gathered.cpu().numpy().view(np.float32).reshape(num_instances, clip.num_tracks, 12)
                expected = ob.oracle_decompress_tracks_batch([clip.blob], np.zeros(num_instances, dtype=np.uint32), times, clip.num_tracks)
                assert np.array_equal(gathered.view(np.uint32), expected.view(np.uint32))
                gather.gathered.zero_()
                torch.cuda.synchronize(device)       # the clear must have happened before the other ranks push again
            dist.barrier()
        gather.close()
        context.unregister_clip(handle)
        context.close()
        open(os.path.join(result_dir, f"peer_ok{rank}"), "w").close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world_size", [2, 4])
def test_peer_gather_between_processes(tmp_path, world_size):
    import torch.multiprocessing as mp
    mp.spawn(_peer_worker, args=(world_size, _free_port(), str(tmp_path)), nprocs=world_size, join=True)
    assert all(os.path.exists(tmp_path / f"peer_ok{rank}") for rank in range(world_size))


def _run_bench(command, env, tmp_path, timeout):
    """runs bench.py; returns (completed process, the ONE stdout line = the compact headline, the full record from bench_details.json)"""
    details_path = os.path.join(str(tmp_path), "bench_details.json")
    completed = subprocess.run(command, cwd=ROOT, env=dict(env, ACLHIP_BENCH_DETAILS=details_path), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=timeout)
    lines = [line for line in completed.stdout.splitlines() if line.startswith("{")]
    assert len(lines) == 1, completed.stdout[-2000:] + completed.stderr[-3000:]
    assert len(lines[0]) < 4096, len(lines[0])          # the driver's parser gave up on round 5's 26 KB line
    headline = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline"):
        assert key in headline, key
    details = json.load(open(details_path))
    assert headline["n_gpus"] == details["n_gpus"] and abs(headline["value"] / details["value"] - 1.0) < 1e-4
    return completed, headline, details


@pytest.mark.parametrize("world_size,workload,instances", [(2, "one_clip", 8192), (8, "cinematic", 1024), (4, "database", 4096)])
def test_bench_dry_run_of_the_multi_gpu_control_flow(world_size, workload, instances, tmp_path):
    """`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N` exactly as the driver launches it, ranks sharing the one
    GPU (ACLHIP_BENCH_BACKEND=gloo): the line must carry the whole-job rate and both gathers, timed separately from the decode."""
    env = dict(os.environ, ACLHIP_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0", ACLHIP_BENCH_SHARDED_INSTANCES="1024")
    command = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world_size}", "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
               os.path.join(ROOT, "bench.py"), "--gpus", str(world_size), "--steps", "20", "--warmup", "5", "--workload", workload, "--instances", str(instances), "--gather", "both"]
    completed, headline, result = _run_bench(command, env, tmp_path, 600)
    assert completed.returncode == 0, completed.stderr[-3000:]
    assert headline["gather"]["p2p_to_rank0_ms"] > 0 and headline["checks"]["communicator_ranks"] == world_size
    assert result["n_gpus"] == world_size and result["scaling"] == "weak" and result["value"] > 0
    assert result["config"]["instances_per_gpu"] == instances
    gather = result["gather"]
    assert "p2p_error" not in gather and "rccl_all_gather_error" not in gather, gather
    assert gather["p2p_to_rank0_ms"] > 0 and gather["rccl_all_gather_ms"] > 0
    assert gather["shard_bytes"] == instances * result["config"]["pose_bytes"]
    assert result["checks"]["communicator_ranks"] == world_size and len(result["checks"]["kernel_ms_per_rank"]) == world_size


def test_bench_launches_itself_under_torchrun(tmp_path):
    """`python bench.py --gpus 2` without WORLD_SIZE (how the driver launches its 1-GPU run) becomes a 2-rank torch.distributed.run"""
    env = {key: value for key, value in os.environ.items() if key not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(ACLHIP_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    command = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "5", "--instances", "4096", "--no-extras", "--gather", "none"]
    completed, headline, result = _run_bench(command, env, tmp_path, 600)
    assert completed.returncode == 0, completed.stderr[-3000:]
    assert headline["n_gpus"] == 2 and headline["checks"]["communicator_ranks"] == 2 and headline["self_check"]["bit_exact"] is True


def test_bench_at_8_ranks_covers_the_8_gpu_configs(tmp_path):
    """The driver's own launch, `torchrun --nproc-per-node 8 bench.py --gpus 8`, as a dry run on the one test GPU (small shards): after
    the headline the line carries BASELINE.json's 8-GPU configs -- the 300-bone rig shards and the database-bound clips whose tiers
    stream in on all ranks together -- each with its whole-job rate and both gathers."""
    world_size = 8
    env = dict(os.environ, ACLHIP_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0", ACLHIP_BENCH_SHARDED_INSTANCES="512")
    command = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world_size}", "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
               os.path.join(ROOT, "bench.py"), "--gpus", str(world_size), "--steps", "20", "--warmup", "5", "--instances", "2048"]
    completed, headline, result = _run_bench(command, env, tmp_path, 900)
    assert completed.returncode == 0, completed.stderr[-3000:]
    assert set(headline["workloads"]) == {"cinematic", "database"} and all(row[0] > 0 for row in headline["workloads"].values()), headline["workloads"]
    assert headline["gather"]["status"] == "done" and len(headline["roofline"]["kernel_ms_per_rank"]) == world_size
    assert result["n_gpus"] == world_size and result["gather"]["status"] == "done"
    # what the line says about the job itself (bench.py: distributed_checks): the communicator counted every rank, every rank reported
    # its device and its kernel time; the ranks of a dry run share one GPU, which the check of distinct devices is not applied to
    checks = result["checks"]
    assert checks["communicator_ranks"] == world_size and checks["n_gpus_claimed"] == world_size and checks["backend"] == "gloo"
    assert len(checks["devices"]) == world_size and len(checks["peer_access"]) == world_size
    assert len(checks["kernel_ms_per_rank"]) == world_size and all(value > 0 for value in checks["kernel_ms_per_rank"])
    assert checks["kernel_ms_min"] == min(checks["kernel_ms_per_rank"]) and checks["kernel_ms_max"] == max(checks["kernel_ms_per_rank"])
    assert result["roofline"]["kernel_ms_per_rank"] == checks["kernel_ms_per_rank"]
    assert not any("ranks, not" in problem or "distinct GPUs" in problem for problem in checks["problems"]), checks
    workloads = {entry["workload"]: entry for entry in result["workloads"]}
    assert set(workloads) == {"cinematic", "database"}, result["workloads"]
    for name, instances in (("cinematic", 512), ("database", 256)):
        entry = workloads[name]
        assert "error" not in entry, entry
        assert entry["n_gpus"] == world_size and entry["instances_per_gpu"] == instances and entry["poses_per_s"] > 0 and entry["kernel_ms"] > 0
        gather = entry["gather"]
        assert "p2p_error" not in gather and "rccl_all_gather_error" not in gather, gather
        assert gather["p2p_to_rank0_ms"] > 0 and gather["rccl_all_gather_ms"] > 0 and gather["shard_bytes"] == instances * entry["pose_bytes"]
    assert workloads["cinematic"]["bones"] == 300
    assert workloads["database"]["database_chunks_streamed_in_together"] > 0


def test_a_gather_that_does_not_come_back_still_leaves_the_decode_line(tmp_path):
    """the first real N > 1 run must report its decode whatever a collective or a peer mapping does on a node this code has never run on:
    with the gather watchdog set to fire at once, the line still arrives -- the communicator's rank count, every rank's kernel time,
    the whole-job rate -- and says that the gather timed out"""
    world_size = 2
    env = dict(os.environ, ACLHIP_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0", ACLHIP_BENCH_GATHER_TIMEOUT="0.0001")
    command = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world_size}", "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
               os.path.join(ROOT, "bench.py"), "--gpus", str(world_size), "--steps", "20", "--warmup", "5", "--instances", "8192", "--no-extras", "--gather", "both"]
    completed, headline, result = _run_bench(command, env, tmp_path, 600)
    assert headline["gather"]["status"] == "timed out" and headline["value"] > 0 and headline["roofline"]["kernel_ms"] > 0
    assert result["gather"]["status"] == "timed out"
    assert result["n_gpus"] == world_size and result["value"] > 0 and result["roofline"]["kernel_ms"] > 0
    checks = result["checks"]
    assert checks["communicator_ranks"] == world_size and len(checks["kernel_ms_per_rank"]) == world_size and all(value > 0 for value in checks["kernel_ms_per_rank"])
    assert result["self_check"]["bit_exact"] is True
