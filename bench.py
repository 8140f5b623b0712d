#!/usr/bin/env python3
"""bench.py -- poses/sec of the batched clip decompressor (BASELINE.json metric), one process per GPU.

    python bench.py --gpus N --steps K --warmup W

A "step" is one launch of aclhip_decompress_tracks_batch (seek + decompress_tracks for every instance of the
batch). Inputs (registered clips, instance lists) and the pose buffer are resident in HBM before the timed region.
Workloads (BASELINE.json configs):
    one_clip   64k instances of one CMU-shaped 100-bone clip, random sample times       (configs[1], the default)
    256_clips  64k instances drawn from 256 distinct 100-bone clips                      (configs[2])
    cinematic  64k instances per GPU of a 300-bone rig with scale, multi-segment         (configs[3], per GPU shard)
    database   64k instances over 16 database-bound 100-bone clips; the low importance tier is streamed in chunk by chunk
               on the decode stream while the batches run                                   (configs[4] shape, committed fixture)
    scalar     64k instances of one 256-curve float1f track list (blend shape weights)      
With N > 1 every rank decodes its own shard of instances (weak scaling, no data-path collective); rank 0 prints ONE
JSON line with the whole-job poses/sec, the roofline of the decode kernel and, at N = 1, the CPU baseline.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0     # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s spec peak (about 6.3 TB/s achievable)
INSTANCES_PER_GPU = 65536


def build_workload(name, rank):
    """Returns (list of SyntheticClip, instance->clip index array, sample times) for this rank's shard."""
    from acl_amd import synth

    rng = np.random.default_rng(1000 + rank)
    if name in ("one_clip", "object_space"):
        clips = [synth.build_clip(seed=2, num_tracks=100, num_samples=301, sample_rate=30.0)]
        clip_indices = np.zeros(INSTANCES_PER_GPU, dtype=np.uint32)
    elif name == "additive_object_space":
        # instance = additive clip 1 applied onto base clip 0 (instance i's base time is drawn in main), then local -> object space
        clips = [synth.build_clip(seed=2, num_tracks=100, num_samples=301, sample_rate=30.0),
                 synth.build_clip(seed=12, num_tracks=100, num_samples=121, sample_rate=30.0, rotation_constant=0.5, translation_constant=0.8)]
        clip_indices = np.ones(INSTANCES_PER_GPU, dtype=np.uint32)
    elif name == "256_clips":
        clips = []
        spec_rng = np.random.default_rng(3)
        for i in range(256):
            animated = spec_rng.uniform(0.25, 0.5)
            clips.append(synth.build_clip(
                seed=300 + i, num_tracks=100, num_samples=int(spec_rng.integers(31, 601)), sample_rate=30.0,
                rotation_default=0.02, rotation_constant=float(0.98 - animated),
                wrap=int(spec_rng.uniform() < 0.1), strip_keyframes=int(spec_rng.uniform() < 0.1),
                min_bits=int(spec_rng.integers(5, 10)), max_bits=int(spec_rng.integers(12, 19))))
        # measurement aid: ACLHIP_BENCH_CLIP_SUBSET=K draws the instances from the first K of the 256 clips only
        subset = int(os.environ.get("ACLHIP_BENCH_CLIP_SUBSET", "256"))
        clip_indices = rng.integers(0, subset, size=INSTANCES_PER_GPU).astype(np.uint32)
    elif name == "cinematic":
        clips = [synth.build_clip(seed=4, num_tracks=300, num_samples=451, sample_rate=30.0, has_scale=1,
                                  scale_default=0.7, scale_constant=0.1, rotation_constant=0.45, translation_constant=0.8)]
        clip_indices = np.zeros(INSTANCES_PER_GPU, dtype=np.uint32)
    elif name == "scalar":
        # 1 % of the curves at the raw bit rate: what the reference's compressor leaves for tracks it cannot quantize within precision
        clips = [synth.build_scalar_clip(seed=9, track_type=0, num_tracks=256, num_samples=120, sample_rate=60.0, raw_fraction=0.01)]
        clip_indices = np.zeros(INSTANCES_PER_GPU, dtype=np.uint32)
    elif name == "database":
        clips = load_database_fixture()["clips"]
        clip_indices = rng.integers(0, len(clips), size=INSTANCES_PER_GPU).astype(np.uint32)
    else:
        raise ValueError(f"unknown workload {name}")

    durations = np.array([c.duration for c in clips], dtype=np.float32)
    times = (rng.uniform(0.0, 1.0, size=clip_indices.size).astype(np.float32) * durations[clip_indices]).astype(np.float32)
    return clips, clip_indices, times


class FixtureClip:
    """A clip that comes from a committed fixture instead of the synthetic writer."""

    def __init__(self, blob):
        from acl_amd import synth
        self.blob = synth.aligned_bytes(blob.size)
        self.blob[:] = blob
        header = np.frombuffer(bytes(self.blob[8:32]), dtype=np.uint32)
        self.num_tracks, self.num_samples = int(header[2]), int(header[3])
        self.sample_rate = float(np.frombuffer(bytes(self.blob[24:28]), dtype=np.float32)[0])
        self.duration = float(np.float32(self.num_samples - 1) / np.float32(self.sample_rate)) if self.num_samples > 1 else 0.0


def load_database_fixture():
    """tests/golden/bench/database_16_clips_100_bones.npz: clips + database + bulk data written by the reference's build_database
    (tests/golden/make_bench_database.py); the reference itself does not exist on the GPU box."""
    from acl_amd import synth
    data = np.load(os.path.join(ROOT, "tests", "golden", "bench", "database_16_clips_100_bones.npz"))
    offsets = data["clip_offsets"]

    def aligned(array):
        out = synth.aligned_bytes(max(array.size, 1))
        out[: array.size] = array
        return out[: array.size]

    return {"clips": [FixtureClip(data["clips"][offsets[i]: offsets[i + 1]]) for i in range(offsets.size - 1)],
            "database": aligned(data["database"]), "bulk_medium": aligned(data["bulk_medium"]), "bulk_low": aligned(data["bulk_low"])}


def cpu_baseline_scalar(clips, clip_indices, times, row_floats):
    """cpu_baseline for scalar track lists: the reference's decoder on all host cores (kind "reference") or the C restatement."""
    import ctypes
    from oracle import bindings as ob  # cpu_baseline leg only

    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    sample = min(clip_indices.size, 65536)
    indices = np.ascontiguousarray(clip_indices[:sample], dtype=np.uint32)
    sample_times = np.ascontiguousarray(times[:sample], dtype=np.float32)
    blob_ptrs = (ctypes.c_void_p * len(clips))(*[c.blob.ctypes.data for c in clips])
    if ob.have_ref_scalar():
        lib = ob.ref_scalar()
        probe = lib.aclref_scalar_bench(blob_ptrs, indices.ctypes.data, sample_times.ctypes.data, min(sample, 8192), row_floats, cores, 1)
        repeats = int(max(1, min(400, 1.0 / max(probe / min(sample, 8192) * sample, 1e-9))))
        seconds = min(lib.aclref_scalar_bench(blob_ptrs, indices.ctypes.data, sample_times.ctypes.data, sample, row_floats, cores, repeats) for _ in range(3))
        return {"value": sample / seconds, "unit": "poses/s", "cores": cores, "kind": "reference",
                "sample": f"{sample} instances of the same list, seek+decompress_tracks, reference headers (default_scalar_decompression_settings), "
                          f"{cores} threads, warm cache, best of 3 rounds of {repeats} passes"}
    sample = min(sample, 4096)
    options = ob.default_options()
    out = np.zeros(row_floats, dtype=np.float32)
    t0 = time.perf_counter()
    for i in range(sample):
        ob.oracle().aclo_scalar_decompress_tracks(clips[indices[i]].blob.ctypes.data, ctypes.c_float(sample_times[i]), 0, ctypes.byref(options), out.ctypes.data)
    seconds = time.perf_counter() - t0
    return {"value": sample / seconds, "unit": "poses/s", "cores": 1, "kind": "port",
            "sample": f"{sample} instances of the same list, scalar C restatement (oracle/acl_oracle.c) called from Python, 1 thread"}


def cpu_baseline(clips, clip_indices, times, max_tracks):
    """Times the reference's own decoder (oracle/_ref, kind "reference") -- or the C restatement (kind "port") when the
    reference build is absent -- on a bounded sample of the same instance list, on this box's host cores."""
    import ctypes
    from oracle import bindings as ob  # cpu_baseline leg: the oracle is the baseline being measured here, never the product

    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    sample = min(clip_indices.size, 65536)
    indices = np.ascontiguousarray(clip_indices[:sample], dtype=np.uint32)
    sample_times = np.ascontiguousarray(times[:sample], dtype=np.float32)

    if ob.have_ref():
        lib = ob.ref()
        blob_ptrs = (ctypes.c_void_p * len(clips))(*[c.blob.ctypes.data for c in clips])
        # calibrate so that the timed part is roughly 15-25 s of CPU work in total (about 2-3 s of wall time on all cores)
        probe = lib.aclref_bench(blob_ptrs, indices.ctypes.data, sample_times.ctypes.data, min(sample, 8192), max_tracks, cores, 1, None)
        per_pose = probe / min(sample, 8192)
        repeats = int(max(1, min(400, 1.0 / max(per_pose * sample, 1e-9))))
        # three rounds (threads are created once per round and walk the list `repeats` times), best round counts
        seconds = min(lib.aclref_bench(blob_ptrs, indices.ctypes.data, sample_times.ctypes.data, sample, max_tracks, cores, repeats, None) for _ in range(3))
        return {"value": sample / seconds, "unit": "poses/s", "cores": cores, "kind": "reference",
                "sample": f"{sample} instances of the same list, seek+decompress_tracks, reference headers (AVX2 build, benchmark settings), "
                          f"{cores} threads, warm cache, best of 3 rounds of {repeats} passes"}

    lib = ob.oracle()
    options = ob.default_options()
    sample = min(sample, 16384)
    out = np.zeros((sample, max_tracks * 12), dtype=np.float32)
    blob_ptrs = (ctypes.c_void_p * len(clips))(*[c.blob.ctypes.data for c in clips])
    best = 1e30
    for _ in range(3):
        t0 = time.perf_counter()
        lib.aclo_decompress_tracks_batch(blob_ptrs, indices.ctypes.data, sample_times.ctypes.data, sample, 0, ctypes.byref(options), out.ctypes.data, max_tracks * 12)
        best = min(best, time.perf_counter() - t0)
    return {"value": sample / best, "unit": "poses/s", "cores": 1, "kind": "port",
            "sample": f"{sample} instances of the same list, scalar C restatement (oracle/acl_oracle.c), 1 thread, best of 3"}


def measured_traffic(workload, kernel_name):
    """HBM bytes per launch of the decode kernel from rocprofv3 PMC passes of this same command, committed under profiles/
    (FETCH_SIZE and WRITE_SIZE collected in separate --pmc runs, KiB units, FETCH_SIZE doubled per the gfx950 note of
    MI355X_MICROARCH.md). None when no committed measurement matches the workload and kernel."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    if not os.path.exists(path):
        return None
    try:
        entries = json.load(open(path))
    except ValueError:
        return None
    for entry in entries:
        if entry.get("workload") == workload and entry.get("kernel") == kernel_name:
            return entry.get("traffic_bytes_per_launch")
    return None


def main():
    parser = argparse.ArgumentParser()
    parser.add_argument("--gpus", type=int, default=1)
    # the clocks of an idle MI355X take a few ms to ramp: the defaults warm up for ~30 ms and time ~0.1 s
    parser.add_argument("--steps", type=int, default=2000)
    parser.add_argument("--warmup", type=int, default=500)
    parser.add_argument("--workload", default="one_clip", choices=["one_clip", "256_clips", "cinematic", "database", "scalar", "object_space", "additive_object_space"])
    parser.add_argument("--sort-by-clip", action="store_true", help="bucket the instance list by clip before upload (256_clips)")
    parser.add_argument("--no-cpu-baseline", action="store_true", help="skip the cpu_baseline leg")
    parser.add_argument("--order-for-locality", action="store_true",
                        help="decode in the order aclhip_order_instances_for_locality gives (every clip on one XCD); the poses are stored in that order")
    parser.add_argument("--keep-rows", action="store_true",
                        help="with --order-for-locality: store every pose in its instance's ORIGINAL row (aclhip_decompress_tracks_batch_rows)")
    args = parser.parse_args()

    import torch
    import torch.distributed as dist

    world_size = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    distributed = world_size > 1
    if args.gpus != world_size and distributed:
        raise SystemExit(f"--gpus {args.gpus} does not match WORLD_SIZE {world_size}")
    if args.gpus > 1 and not distributed:
        raise SystemExit("launch N > 1 through torch.distributed.run (one process per GPU)")

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU")
    # ACLHIP_BENCH_BACKEND=gloo is a dry run of the N > 1 control flow on a box with fewer GPUs than ranks (ranks then share GPUs);
    # the real thing is one rank per GPU over RCCL (backend "nccl")
    backend = os.environ.get("ACLHIP_BENCH_BACKEND", "nccl")
    device_index = local_rank if backend == "nccl" else local_rank % torch.cuda.device_count()
    torch.cuda.set_device(device_index)
    device = torch.device("cuda", device_index)
    if distributed:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=device)
        else:
            dist.init_process_group(backend=backend)

    from acl_amd import runtime

    clips, clip_indices, times = build_workload(args.workload, rank)
    if args.sort_by_clip:
        order = np.argsort(clip_indices, kind="stable")
        clip_indices, times = clip_indices[order], times[order]

    context = runtime.Context(device_index)
    registration_t0 = time.perf_counter()
    is_scalar = args.workload == "scalar"
    database = None
    if args.workload == "database":
        fixture = load_database_fixture()
        database = context.register_database(fixture["database"], fixture["bulk_medium"] if fixture["bulk_medium"].size else None,
                                             fixture["bulk_low"] if fixture["bulk_low"].size else None)
        handles = np.array([context.register_clip_with_database(c.blob, database) for c in clips], dtype=np.uint32)
    else:
        handles = np.array([context.register_clip(c.blob) for c in clips], dtype=np.uint32)
    registration_ms = (time.perf_counter() - registration_t0) * 1e3
    max_tracks = max(c.num_tracks for c in clips)
    num_instances = clip_indices.size
    # bytes of one instance's output row: 48 per transform track (rtm::qvvf), 4 per component of a scalar track
    pose_stride = max_tracks * (4 * clips[0].num_components if is_scalar else 48)

    # --order-for-locality: the library lays the instance list out (host side, setup), the poses still land in the caller's rows
    d_rows = None
    ordering_ms = None
    if args.order_for_locality:
        if is_scalar or args.workload in ("object_space", "additive_object_space"):
            raise SystemExit("--order-for-locality applies to the pose kernels")
        ordering_t0 = time.perf_counter()
        order = context.order_instances_for_locality(handles[clip_indices])
        ordering_ms = (time.perf_counter() - ordering_t0) * 1e3
        clip_indices, times = clip_indices[order], times[order]
        if args.keep_rows:
            d_rows = torch.from_numpy(order.astype(np.int32)).to(device)

    d_clips = torch.from_numpy(handles[clip_indices].astype(np.int32)).to(device)
    d_times = torch.from_numpy(times).to(device)
    d_poses = torch.empty((num_instances, pose_stride // 4), dtype=torch.float32, device=device)
    stream = torch.cuda.current_stream(device)
    params = runtime.default_params()

    # raw pointers once: the timed loop is nothing but K asynchronous launches through the C ABI
    import ctypes
    lib = runtime.load_library()
    launch_args = (context._handle, d_clips.data_ptr(), d_times.data_ptr(), num_instances, ctypes.byref(params), d_poses.data_ptr(), pose_stride, stream.cuda_stream)
    launch = lib.aclhip_decompress_scalar_tracks_batch if is_scalar else lib.aclhip_decompress_tracks_batch
    if d_rows is not None:
        launch_args = (context._handle, d_clips.data_ptr(), d_times.data_ptr(), d_rows.data_ptr(), num_instances, ctypes.byref(params), d_poses.data_ptr(), pose_stride, stream.cuda_stream)
        launch = lib.aclhip_decompress_tracks_batch_rows

    # pose consumers: the same decode with the additive apply / local -> object space fused in
    consumers = None
    if args.workload in ("object_space", "additive_object_space"):
        from acl_amd import synth
        parents = synth.humanoid_hierarchy(max_tracks)     # 13 depths, 4-18 transforms wide
        for handle in handles:
            context.set_clip_hierarchy(int(handle), parents)
        consumers = runtime.PoseConsumers()
        consumers.object_space = 1
        if args.workload == "additive_object_space":
            base_rng = np.random.default_rng(2000 + rank)
            d_base_clips = torch.full((num_instances,), int(handles[0]), dtype=torch.int32, device=device)
            d_base_times = torch.from_numpy(base_rng.uniform(0.0, clips[0].duration, size=num_instances).astype(np.float32)).to(device)
            consumers.additive_format = runtime.ADDITIVE_ADDITIVE1
            consumers.base_clips = d_base_clips.data_ptr()
            consumers.base_sample_times = d_base_times.data_ptr()
        launch_args = (context._handle, d_clips.data_ptr(), d_times.data_ptr(), num_instances, ctypes.byref(params), ctypes.byref(consumers), d_poses.data_ptr(), pose_stride, stream.cuda_stream)
        launch = lib.aclhip_decompress_poses_batch

    def step():
        status = launch(*launch_args)
        if status != 0:
            raise SystemExit(f"the batch launch failed: {status}")

    # device pre-warm (setup, not one of the W warm-up steps): an idle MI355X needs a few ms of work before its clocks settle
    # (skipped under a counter-collecting profiler, where every launch is serialized and slow: ACLHIP_BENCH_PROFILING=1)
    profiling = os.environ.get("ACLHIP_BENCH_PROFILING", "0") == "1"
    prewarm_deadline = time.perf_counter() + (0.0 if profiling else 0.15)
    while time.perf_counter() < prewarm_deadline:
        for _ in range(64):
            step()
        torch.cuda.synchronize(device)

    for _ in range(args.warmup):
        step()

    # HIP events on the launch stream cut the timed region into chunks of launches; (chunk time / launches in it) averaged over the
    # region is the kernel's mean duration including the (~1 us) launch boundary
    chunk = max(1, min(100, args.steps // 10))
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps // chunk + 2)]

    if distributed:
        dist.barrier()
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    num_marks = 0
    # database workload: the tiers arrive one chunk at a time, evenly spread over the timed steps, on the decode stream
    stream_schedule = {}
    if database is not None:
        info = context.database_info(database)
        pending = [(tier, 1) for tier in (runtime.TIER_MEDIUM_IMPORTANCE, runtime.TIER_LOWEST_IMPORTANCE) for _ in range(info.num_chunks[tier - 1])]
        for k, request in enumerate(pending):
            stream_schedule[(k + 1) * args.steps // (len(pending) + 1)] = request
    for i in range(args.steps):
        if i % chunk == 0:
            marks[num_marks].record(stream)
            num_marks += 1
        if i in stream_schedule:
            context.database_stream_in(database, stream_schedule[i][0], stream_schedule[i][1], stream=stream.cuda_stream)
        step()
    marks[num_marks].record(stream)
    num_marks += 1
    torch.cuda.synchronize(device)
    if distributed:
        dist.barrier()
    elapsed = time.perf_counter() - t0

    if distributed:
        elapsed_tensor = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(elapsed_tensor, op=dist.ReduceOp.MAX)
        elapsed = float(elapsed_tensor.item())

    # Roofline of the decode kernel: device time per launch from the HIP events of the timed region
    kernel_ms = float(marks[0].elapsed_time(marks[num_marks - 1])) / args.steps
    # the same launches back to back from C (no host pacing), for reference
    repeats = 1 if profiling else max(10, min(args.steps, 100))
    if is_scalar:
        kernel_ms_back_to_back = None
    elif consumers is not None:
        kernel_ms_back_to_back = context.time_decompress_poses_batch(d_clips.data_ptr(), d_times.data_ptr(), num_instances, d_poses.data_ptr(), pose_stride,
                                                                     consumers, repeats=repeats, params=params, stream=stream.cuda_stream)
    else:
        kernel_ms_back_to_back = context.time_decompress_tracks_batch(d_clips.data_ptr(), d_times.data_ptr(), num_instances, d_poses.data_ptr(), pose_stride,
                                                                      repeats=repeats, params=params, stream=stream.cuda_stream)
    bytes_written, bytes_read = context.batch_algorithmic_bytes(handles[clip_indices])
    if args.workload == "additive_object_space":
        bytes_read += context.batch_algorithmic_bytes(handles[:1])[1]        # the base clip is read too; one pose per instance is written
    algorithmic_bytes = bytes_written + bytes_read
    achieved_gbps = algorithmic_bytes / (kernel_ms * 1e-3) / 1e9

    write_ceiling_gbps = context.measure_write_bandwidth(d_poses.data_ptr(), num_instances * pose_stride, repeats=1 if profiling else 20, stream=stream.cuda_stream)

    rejected = context.rejected_instance_count()
    if rejected != 0:
        raise SystemExit(f"the kernel rejected {rejected} instances")

    # committed PMC measurements are per instance order: random (the default), or the library's locality order
    traffic_key = args.workload
    if args.order_for_locality:
        traffic_key = None if args.keep_rows else args.workload + ", aclhip_order_instances_for_locality order"
    elif args.sort_by_clip:
        traffic_key = None
    kernel_name = "decompress_scalar_tracks_kernel" if is_scalar else ("decompress_poses_consumer_kernel" if consumers is not None else context.tracks_kernel_name(params))
    if rank == 0:
        total_poses = num_instances * world_size * args.steps
        result = {
            "metric": "poses/sec (whole node), 64k clip instances x 100 bones per GPU, seek + decompress_tracks",
            "value": total_poses / elapsed,
            "unit": "poses/s",
            "n_gpus": world_size,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": {"one_clip": "64k instances of one CMU-shaped 100-bone clip, random sample times, quatf_drop_w_variable + vector3f_variable (BASELINE.json configs[1])",
                             "256_clips": "64k instances drawn from 256 distinct 100-bone clips (BASELINE.json configs[2])" + (", bucketed by clip" if args.sort_by_clip else "") + (", decoded in aclhip_order_instances_for_locality order" + (", poses scattered back to their original rows" if args.keep_rows else "") if args.order_for_locality else ""),
                             "cinematic": "64k instances per GPU of a 300-bone rig with scale tracks, multi-segment (BASELINE.json configs[3] shard)",
                             "database": "64k instances per GPU over 16 database-bound 100-bone clips, low importance tier streamed in chunk by chunk on the decode stream during the timed steps (BASELINE.json configs[4] shape, committed fixture)",
                             "scalar": "64k instances per GPU of one 256-curve float1f track list (scalar tracks, SURVEY 8 f4)",
                             "object_space": "the one_clip batch with local -> object space fused into the decode (pose consumers, SURVEY 8 f3)",
                             "additive_object_space": "64k instances per GPU: an additive clip applied (additive1) onto a base clip instance decoded by the same wave, then local -> object space (SURVEY 8 f3)"}[args.workload],
                "instances_per_gpu": int(num_instances),
                "bones": int(max_tracks),
                "distinct_clips": len(clips),
                "ordering_ms": None if ordering_ms is None else round(ordering_ms, 3),     # aclhip_order_instances_for_locality on the host (setup, not timed)
                "registration_ms_total": round(registration_ms, 3),      # validate + derive tables + upload, all clips (setup, not timed)
                "pose_bytes": int(pose_stride),
                "sharding": f"instances split over {world_size} rank(s), no collective on the data path",
            },
            "roofline": {
                "bound": "hbm",
                "achieved": achieved_gbps,
                "peak": HBM_PEAK_GBPS,
                "unit": "GB/s",
                "frac": achieved_gbps / HBM_PEAK_GBPS,
                "traffic": measured_traffic(traffic_key, kernel_name),
                "kernel": kernel_name,
                "kernel_ms": kernel_ms,
                "kernel_ms_back_to_back": kernel_ms_back_to_back,
                "algorithmic_bytes_per_launch": int(algorithmic_bytes),
                "measured_write_stream_gbps": write_ceiling_gbps,
            },
        }
        if world_size == 1 and not args.no_cpu_baseline:
            if is_scalar:
                result["cpu_baseline"] = cpu_baseline_scalar(clips, clip_indices, times, pose_stride // 4)
            else:
                result["cpu_baseline"] = cpu_baseline(clips, clip_indices, times, max_tracks)
                if database is not None:
                    # the reference's database_context is not part of the CPU bridge that travels to the GPU box
                    result["cpu_baseline"]["sample"] += "; clips bound WITHOUT their database (highest importance tier only)"
                if consumers is not None:
                    result["cpu_baseline"]["sample"] += "; decode of the (additive) clip only, the consumers are not part of the CPU timing"
        print(json.dumps(result))

    for handle in handles:
        context.unregister_clip(int(handle))
    if database is not None:
        context.unregister_database(database)
    context.close()
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
