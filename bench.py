#!/usr/bin/env python3
"""bench.py -- poses/sec of the batched clip decompressor (BASELINE.json metric), one process per GPU.

    python bench.py --gpus N --steps K --warmup W

A "step" is one launch of aclhip_decompress_tracks_batch (seek + decompress_tracks for every instance of the
batch). Inputs (registered clips, instance lists) and the pose buffer are resident in HBM before the timed region.
Workloads (BASELINE.json configs):
    one_clip   64k instances of one CMU-shaped 100-bone clip, random sample times       (configs[1], the default = the headline)
    256_clips  64k instances drawn from 256 distinct 100-bone clips                      (configs[2])
    cinematic  64k instances per GPU of a 300-bone rig with scale, multi-segment         (configs[3], per GPU shard)
    database   64 database-bound 100-bone clips (medium 0 % / low 50 %), the low importance tier is streamed in chunk by chunk
               on the decode stream while the batches run                                   (configs[4], committed fixture)
    scalar     64k instances of one 256-curve float1f track list (blend shape weights)      
    object_space / additive_object_space   the pose consumers fused into the decode         
With N > 1 every rank decodes its own shard of instances (weak scaling, no data-path collective); rank 0 prints ONE
JSON line (< 4 KB: compact_headline) with the whole-job poses/sec, the roofline of the decode kernel, the CPU baseline and one row per
extra workload; the full record of the run goes to bench_details.json next to this file and to stderr. `python bench.py --gpus N`
without WORLD_SIZE re-executes itself under torch.distributed.run with N ranks. At N = 1 the default run also measures, in the same
process and OUTSIDE the timed region, the other north-star configs ("workloads"), a footprint sweep of the headline batch, the
compact output layouts, and the CPU baseline (a pinned thread sweep of the reference's own decoder). At N > 1 the pose gather
(RCCL all-gather, peer-to-peer writes into rank 0) is timed separately from the decode ("gather").
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0     # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s spec peak (about 6.3 TB/s achievable)
XGMI_LINK_GBPS = 153.0     # one xGMI link, one direction (7 links per GPU)
INSTANCES_PER_GPU = 65536

WORKLOAD_TEXT = {
    "one_clip": "64k instances of one CMU-shaped 100-bone clip, random sample times, quatf_drop_w_variable + vector3f_variable (BASELINE.json configs[1])",
    "256_clips": "64k instances drawn from 256 distinct 100-bone clips (BASELINE.json configs[2])",
    "cinematic": "64k instances per GPU of a 300-bone rig with scale tracks, multi-segment (BASELINE.json configs[3] shard)",
    "database": "instances over 64 database-bound 100-bone clips (medium 0 % / low 50 %), low importance tier streamed in chunk by chunk on the decode stream "
                "during the timed steps (BASELINE.json configs[4], committed fixture)",
    "scalar": "64k instances per GPU of one 256-curve float1f track list (scalar tracks, SURVEY 8 f4)",
    "object_space": "the one_clip batch with local -> object space fused into the decode (pose consumers, SURVEY 8 f3)",
    "additive_object_space": "64k instances per GPU: an additive clip applied (additive1) onto a base clip instance decoded by the same wave, then local -> object space (SURVEY 8 f3)",
    "object_space_fast": "object_space with ACLHIP_CONSUMERS_FAST: the opt-in 1 ulp arithmetic (poses within 2e-6 of the bit exact kernels')",
    "additive_object_space_fast": "additive_object_space with ACLHIP_CONSUMERS_FAST",
    "blend_object_space": "64k instances per GPU, each the weighted blend of three clip instances (three 100-bone clips of one skeleton), then local -> object space (SURVEY 8 f3)",
    "cinematic_16": "64k instances per GPU drawn from 16 distinct 300-bone rigs with scale (measurement aid: poses of several windows over several clips)",
    "one_clip_mixed_registry": "the one_clip batch (4 800 byte rows) while the context ALSO holds a 300-bone rig and a 551-bone clip: the launch is shaped by the batch, not by the registry",
    "track_requests": "4 M random (instance, bone) requests on the 100-bone clip: seek + decompress_track, one 48 byte qvv per request (SURVEY 8 a15)",
    "track_requests_256_clips": "4 M random (instance, bone) requests, every request's clip drawn from 256 distinct 100-bone clips: waves of mixed clips (--order locality: the same requests in aclhip_order_track_requests_for_locality order; --order by_clip: sorted by clip)",
    "one_clip_lods": "the one_clip batch with a per character LOD: every instance stores its first 100 / 60 / 30 bones (a third of the crowd each, "
                     "aclhip_output_desc::instance_track_counts): one launch, 63 % of the bytes",
}
LOD_TRACK_COUNTS = (100, 60, 30)
TRACK_REQUESTS = 1 << 22


_workload_cache = {}


def build_workload(name, rank, num_instances):
    """Returns (list of SyntheticClip, instance->clip index array, sample times) for this rank's shard (the same objects when asked again)."""
    key = (name, rank, num_instances, os.environ.get("ACLHIP_BENCH_CLIP_SUBSET"))
    if key not in _workload_cache:
        clips, clip_indices, times = _build_workload(name, rank, num_instances)
        _workload_cache[key] = (clips, clip_indices, times)
    clips, clip_indices, times = _workload_cache[key]
    return clips, clip_indices.copy(), times.copy()


def _build_workload(name, rank, num_instances):
    from acl_amd import synth

    rng = np.random.default_rng(1000 + rank)
    if name in ("one_clip", "object_space", "object_space_fast", "one_clip_mixed_registry", "track_requests", "one_clip_lods"):
        clips = [synth.build_clip(seed=2, num_tracks=100, num_samples=301, sample_rate=30.0)]
        clip_indices = np.zeros(num_instances, dtype=np.uint32)
    elif name == "blend_object_space":
        # three clips of one skeleton (a locomotion blend): instance i blends clip_indices[i] with two more drawn in Job
        clips = [synth.build_clip(seed=2, num_tracks=100, num_samples=301, sample_rate=30.0),
                 synth.build_clip(seed=22, num_tracks=100, num_samples=241, sample_rate=30.0),
                 synth.build_clip(seed=23, num_tracks=100, num_samples=181, sample_rate=30.0)]
        clip_indices = np.zeros(num_instances, dtype=np.uint32)
    elif name in ("additive_object_space", "additive_object_space_fast"):
        # instance = additive clip 1 applied onto base clip 0 (instance i's base time is drawn in Job), then local -> object space
        clips = [synth.build_clip(seed=2, num_tracks=100, num_samples=301, sample_rate=30.0),
                 synth.build_clip(seed=12, num_tracks=100, num_samples=121, sample_rate=30.0, rotation_constant=0.5, translation_constant=0.8)]
        clip_indices = np.ones(num_instances, dtype=np.uint32)
    elif name in ("256_clips", "track_requests_256_clips"):
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
        clip_indices = rng.integers(0, subset, size=num_instances).astype(np.uint32)
    elif name == "cinematic":
        clips = [synth.build_clip(seed=4, num_tracks=300, num_samples=451, sample_rate=30.0, has_scale=1,
                                  scale_default=0.7, scale_constant=0.1, rotation_constant=0.45, translation_constant=0.8)]
        clip_indices = np.zeros(num_instances, dtype=np.uint32)
    elif name == "cinematic_16":
        clips = [synth.build_clip(seed=40 + i, num_tracks=300, num_samples=200 + 17 * i, sample_rate=30.0, has_scale=1,
                                  scale_default=0.7, scale_constant=0.1, rotation_constant=0.45, translation_constant=0.8) for i in range(16)]
        clip_indices = rng.integers(0, 16, size=num_instances).astype(np.uint32)
    elif name == "scalar":
        # 1 % of the curves at the raw bit rate: what the reference's compressor leaves for tracks it cannot quantize within precision
        clips = [synth.build_scalar_clip(seed=9, track_type=0, num_tracks=256, num_samples=120, sample_rate=60.0, raw_fraction=0.01)]
        clip_indices = np.zeros(num_instances, dtype=np.uint32)
    elif name == "database":
        clips = load_database_fixture()["clips"]
        clip_indices = rng.integers(0, len(clips), size=num_instances).astype(np.uint32)
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
        self.num_components = 12


_database_fixture = None


def load_database_fixture():
    """tests/golden/bench/database_64_clips_100_bones.npz: clips + database + bulk data written by the reference's build_database
    (tests/golden/make_bench_database.py); the reference itself does not exist on the GPU box."""
    global _database_fixture
    if _database_fixture is None:
        from acl_amd import synth
        data = np.load(os.path.join(ROOT, "tests", "golden", "bench", "database_64_clips_100_bones.npz"))
        offsets = data["clip_offsets"]

        def aligned(array):
            out = synth.aligned_bytes(max(array.size, 1))
            out[: array.size] = array
            return out[: array.size]

        _database_fixture = {"clips": [FixtureClip(data["clips"][offsets[i]: offsets[i + 1]]) for i in range(offsets.size - 1)],
                             "database": aligned(data["database"]), "bulk_medium": aligned(data["bulk_medium"]), "bulk_low": aligned(data["bulk_low"])}
    return _database_fixture


# ---- one workload resident on one GPU ---------------------------------------------------------------------------------------

class Job:
    """Registered clips, instance list and pose buffer of one workload in HBM, and the launch through the C ABI.
    order: "random" (as drawn), "by_clip" (host bucketed), "locality" (aclhip_order_instances_for_locality on the host, setup),
    "device" (aclhip_order_instances_device in front of EVERY launch, on the launch stream: the ordering is part of the step),
    "list" (a persistent aclhip_instance_list: ordered once at setup; EVERY step 1 % of the instances change clip through
    aclhip_instance_list_update and the list is decoded with aclhip_decompress_tracks_list -- the library re-orders it when an
    eighth of it has changed: update, decode and the re-orders that fall into the timed steps are all part of the step),
    "attached" (the same frame loop with aclhip_instance_list_attach: the list decodes the CALLER's clip array; EVERY step a caller side
    kernel -- torch's index_copy_, standing in for the animation graph -- writes the 1 % changes into that array on the decode stream,
    aclhip_instance_list_note_changes counts them, the list is decoded: the caller's kernel, the decode and the re-orders are all part of the step).
    layout: output layout name of runtime.LAYOUTS ("qvv48" = rtm::qvvf records, the default)."""

    def __init__(self, name, rank, device_index, num_instances=INSTANCES_PER_GPU, order="random", keep_rows=False, layout="qvv48", paging="decode_stream", fast=False):
        import torch
        from acl_amd import runtime, synth

        self.name, self.order, self.layout, self.keep_rows, self.fast = name, order, layout, keep_rows, fast
        self.torch, self.runtime = torch, runtime
        self.device = torch.device("cuda", device_index)
        self.lib = runtime.load_library()
        self.clips, clip_indices, times = build_workload(name, rank, num_instances)
        self.is_scalar = name == "scalar"
        self.context = runtime.Context(device_index)
        context = self.context

        t0 = time.perf_counter()
        self.database = None
        if name == "database":
            fixture = load_database_fixture()
            self.database = context.register_database(fixture["database"], fixture["bulk_medium"] if fixture["bulk_medium"].size else None,
                                                      fixture["bulk_low"] if fixture["bulk_low"].size else None)
            self.handles = np.array([context.register_clip_with_database(c.blob, self.database) for c in self.clips], dtype=np.uint32)
        else:
            self.handles = np.array([context.register_clip(c.blob) for c in self.clips], dtype=np.uint32)
        self.registration_ms = (time.perf_counter() - t0) * 1e3
        self.bystanders = []
        if name == "one_clip_mixed_registry":
            # clips the batch never names: the configs[3] rig and a 551-bone crowd leader (docs/fight_scene_performance.md:19-22 of the reference)
            for bystander in (synth.build_clip(seed=4, num_tracks=300, num_samples=451, sample_rate=30.0, has_scale=1, scale_default=0.7, scale_constant=0.1, rotation_constant=0.45, translation_constant=0.8),
                              synth.build_clip(seed=5, num_tracks=551, num_samples=61, sample_rate=30.0)):
                self.bystanders.append(context.register_clip(bystander.blob))

        self.max_tracks = max(c.num_tracks for c in self.clips)
        self.num_instances = int(clip_indices.size)
        self.track_requests = name in ("track_requests", "track_requests_256_clips")
        # bytes of one instance's output row: 48 / 40 / 32 per transform track by layout, 4 per component of a scalar track
        bytes_per_track = 4 * self.clips[0].num_components if self.is_scalar else runtime.LAYOUTS[layout][1]
        # rows start on 64 byte HBM access granules: a stride that is only a multiple of 16 (QVV40: 4000 bytes for 100 bones) leaves every
        # other pose's 1 KiB stores straddling granules (measured: 82 us instead of 44 us); 4800 / 14400 / 3200 already are multiples of 64
        self.pose_stride = 48 if self.track_requests else (self.max_tracks * bytes_per_track + 63) // 64 * 64

        self.ordering_ms = None
        self.d_rows = None
        if order == "by_clip":
            permutation = np.argsort(clip_indices, kind="stable")
            clip_indices, times = clip_indices[permutation], times[permutation]
        elif order == "locality":
            t0 = time.perf_counter()
            # (single track requests: 256 of them per workgroup -- aclhip_order_track_requests_for_locality; poses: aclhip_order_instances_for_locality)
            permutation = runtime.order_track_requests_for_locality(self.handles[clip_indices]) if self.track_requests else context.order_instances_for_locality(self.handles[clip_indices])
            self.ordering_ms = (time.perf_counter() - t0) * 1e3
            clip_indices, times = clip_indices[permutation], times[permutation]
            if keep_rows:
                self.d_rows = torch.from_numpy(permutation.astype(np.int32)).to(self.device)
        self.clip_indices, self.times = clip_indices, times

        self.d_clips = torch.from_numpy(self.handles[clip_indices].astype(np.int32)).to(self.device)
        self.d_times = torch.from_numpy(times).to(self.device)
        self._order_args = None
        if order in ("device", "device_pipelined"):
            # the caller's lists stay as drawn; every step orders them into the lists the decode reads
            self.d_source_clips, self.d_source_times = self.d_clips.clone(), self.d_times.clone()
            self.d_order = torch.zeros((self.num_instances,), dtype=torch.int32, device=self.device)
            if keep_rows:
                self.d_rows = self.d_order
        self.d_poses = torch.empty((self.num_instances, self.pose_stride // 4), dtype=torch.float32, device=self.device)
        self.stream = torch.cuda.current_stream(self.device)
        # database workload: the stream the tiers are paged in on -- the decode stream (every copy between two launches) or a second one
        # (copies overlap the launches: database_streamer.h:87-93 lets a stream-in run concurrently with decompression)
        self.paging = paging
        self.paging_stream = torch.cuda.Stream(self.device) if paging == "second_stream" else self.stream
        # fast: ACLHIP_DECODE_FAST (opt in: rotations within 2e-6 of the bit exact kernels', everything else bit identical)
        self.params = runtime.default_params(flags=runtime.DECODE_FAST if fast else 0)
        self.consumers = None
        self.output = None

        handle, n = context._handle, self.num_instances
        clips_ptr, times_ptr, poses_ptr, stream_ptr = self.d_clips.data_ptr(), self.d_times.data_ptr(), self.d_poses.data_ptr(), self.stream.cuda_stream
        if self.is_scalar:
            self._launch, self._args = self.lib.aclhip_decompress_scalar_tracks_batch, (handle, clips_ptr, times_ptr, n, ctypes.byref(self.params), poses_ptr, self.pose_stride, stream_ptr)
        elif self.track_requests:
            # one request = (instance, bone): random bones of random instances, the reference's decompress_bone use (docs/decompression_performance.md)
            track_rng = np.random.default_rng(4000 + rank)
            self.d_tracks = torch.from_numpy(track_rng.integers(0, self.max_tracks, size=n).astype(np.int32)).to(self.device)
            self._launch = self.lib.aclhip_decompress_track_batch
            self._args = (handle, clips_ptr, times_ptr, self.d_tracks.data_ptr(), n, ctypes.byref(self.params), poses_ptr, stream_ptr)
        elif name in ("object_space", "additive_object_space", "blend_object_space", "object_space_fast", "additive_object_space_fast"):
            parents = synth.humanoid_hierarchy(self.max_tracks)     # 13 depths, 4-18 transforms wide
            for clip_handle in self.handles:
                context.set_clip_hierarchy(int(clip_handle), parents)
            self.consumers = runtime.PoseConsumers()
            self.consumers.object_space = 1
            self.consumers.flags = runtime.CONSUMERS_FAST if name.endswith("_fast") else 0
            if name.startswith("additive_object_space"):
                base_rng = np.random.default_rng(2000 + rank)
                self.d_base_clips = torch.full((n,), int(self.handles[0]), dtype=torch.int32, device=self.device)
                self.d_base_times = torch.from_numpy(base_rng.uniform(0.0, self.clips[0].duration, size=n).astype(np.float32)).to(self.device)
                self.consumers.additive_format = runtime.ADDITIVE_ADDITIVE1
                self.consumers.base_clips = self.d_base_clips.data_ptr()
                self.consumers.base_sample_times = self.d_base_times.data_ptr()
            if name == "blend_object_space":
                blend_rng = np.random.default_rng(5000 + rank)
                others = np.stack([np.full(n, 1), np.full(n, 2)], axis=1)
                self.blend_others = others
                other_times = np.stack([blend_rng.uniform(0.0, self.clips[k].duration, size=n) for k in (1, 2)], axis=1).astype(np.float32)
                weights = blend_rng.dirichlet(np.ones(3), size=n).astype(np.float32)
                self.d_blend_clips = torch.from_numpy(self.handles[others].astype(np.int32)).to(self.device)
                self.d_blend_times = torch.from_numpy(other_times).to(self.device)
                self.d_blend_weights = torch.from_numpy(weights).to(self.device)
                self.consumers.num_blend_clips = 3
                self.consumers.blend_clips, self.consumers.blend_sample_times, self.consumers.blend_weights = self.d_blend_clips.data_ptr(), self.d_blend_times.data_ptr(), self.d_blend_weights.data_ptr()
            self._launch = self.lib.aclhip_decompress_poses_batch
            self._args = (handle, clips_ptr, times_ptr, n, ctypes.byref(self.params), ctypes.byref(self.consumers), poses_ptr, self.pose_stride, stream_ptr)
        elif layout != "qvv48" or self.d_rows is not None or name == "one_clip_lods":
            self.output = runtime.OutputDesc()
            self.output.layout = runtime.LAYOUTS[layout][0]
            if self.d_rows is not None:
                self.output.rows = self.d_rows.data_ptr()
            if name == "one_clip_lods":
                # the writer of every pose keeps its first K bones (track_writer::skip_track_*(track_index), core/track_writer.h:189-191, per character)
                lod_rng = np.random.default_rng(6000 + rank)
                self.track_counts = lod_rng.choice(np.array(LOD_TRACK_COUNTS, dtype=np.int32), size=n)
                self.d_track_counts = torch.from_numpy(self.track_counts).to(self.device)
                self.output.instance_track_counts = self.d_track_counts.data_ptr()
            self._launch = self.lib.aclhip_decompress_tracks_batch_out
            self._args = (handle, clips_ptr, times_ptr, n, ctypes.byref(self.params), ctypes.byref(self.output), poses_ptr, self.pose_stride, stream_ptr)
        else:
            self._launch, self._args = self.lib.aclhip_decompress_tracks_batch, (handle, clips_ptr, times_ptr, n, ctypes.byref(self.params), poses_ptr, self.pose_stride, stream_ptr)

        if order == "device":
            self._order_args = (handle, self.d_source_clips.data_ptr(), self.d_source_times.data_ptr(), n, self.d_order.data_ptr(), clips_ptr, times_ptr, stream_ptr)
        self._pipeline = None
        if order == "device_pipelined":
            # a frame loop knows next frame's clip list before this frame's poses are consumed: the ordering of step k + 1 runs on a SECOND
            # stream while step k decodes -- two sets of ordered lists, an event each way (ordered -> decode may start; decoded -> the set
            # may be overwritten)
            # (ACLHIP_BENCH_ORDER_PRIORITY=1: the ordering's stream at high priority, so that its workgroups are dispatched ahead of the decode's)
            self.order_stream = torch.cuda.Stream(self.device, priority=-1 if os.environ.get("ACLHIP_BENCH_ORDER_PRIORITY", "0") == "1" else 0)
            sets = []
            for _ in range(2):
                ordered_clips, ordered_times, order_out = torch.zeros_like(self.d_clips), torch.zeros_like(self.d_times), torch.zeros_like(self.d_order)
                sets.append({"clips": ordered_clips, "times": ordered_times, "order": order_out, "ordered": torch.cuda.Event(), "decoded": torch.cuda.Event(),
                             "order_args": (handle, self.d_source_clips.data_ptr(), self.d_source_times.data_ptr(), n, order_out.data_ptr(), ordered_clips.data_ptr(), ordered_times.data_ptr(), self.order_stream.cuda_stream),
                             "decode_args": (handle, ordered_clips.data_ptr(), ordered_times.data_ptr(), n, ctypes.byref(self.params), poses_ptr, self.pose_stride, stream_ptr)})
            self._pipeline = {"sets": sets, "step": 0}
            self._enqueue_ordering(sets[0])                    # the first step's list (setup)
            for item in sets:
                item["decoded"].record(self.stream)
        self.instance_list = None
        if order in ("list", "attached"):
            # 16 pre-drawn update sets (1 % of the instances each, new clips drawn like the old ones), cycled through by the steps
            update_rng = np.random.default_rng(3000 + rank)
            count = max(1, n // 100)
            self._updates = []
            for _ in range(16):
                instances = update_rng.choice(n, size=count, replace=False).astype(np.int32)
                new_clips = self.handles[update_rng.integers(0, len(self.clips), size=count)].astype(np.int32)
                self._updates.append((torch.from_numpy(instances).to(self.device), torch.from_numpy(new_clips).to(self.device), count))
            self._update_index = 0
            self.instance_list = context.instance_list_create(n)
            if order == "attached":
                self._updates = [(instances.long(), new_clips, count) for instances, new_clips, count in self._updates]       # (index_copy_ takes 64 bit indices)
                context.instance_list_attach(self.instance_list, clips_ptr, stream=stream_ptr)
            else:
                context.instance_list_set_clips(self.instance_list, clips_ptr, stream=stream_ptr)
            self.stream.synchronize()

    def _enqueue_ordering(self, item):
        self.order_stream.wait_event(item["decoded"])          # the decode that read this set last has finished
        status = self.lib.aclhip_order_instances_device(*item["order_args"])
        if status != 0:
            raise SystemExit(f"the device side ordering failed: {status} {self.lib.aclhip_last_error_message(self.context._handle).decode()}")
        item["ordered"].record(self.order_stream)

    def order_step(self):
        status = self.lib.aclhip_order_instances_device(*self._order_args)
        if status != 0:
            raise SystemExit(f"the device side ordering failed: {status} {self.lib.aclhip_last_error_message(self.context._handle).decode()}")

    def step(self):
        if self.instance_list is not None:
            instances, new_clips, count = self._updates[self._update_index % len(self._updates)]
            self._update_index += 1
            if self.order == "attached":
                self.caller_update(instances, new_clips)
                self.context.instance_list_note_changes(self.instance_list, count)
            else:
                self.context.instance_list_update(self.instance_list, instances.data_ptr(), new_clips.data_ptr(), count, stream=self.stream.cuda_stream)
            self.context.decompress_tracks_list(self.instance_list, self.d_times.data_ptr(), self.d_poses.data_ptr(), self.pose_stride, params=self.params, stream=self.stream.cuda_stream)
            return
        if self._pipeline is not None:
            sets, k = self._pipeline["sets"], self._pipeline["step"]
            self._pipeline["step"] = k + 1
            self._enqueue_ordering(sets[(k + 1) % 2])           # next step's list, on the second stream
            current = sets[k % 2]
            self.stream.wait_event(current["ordered"])
            status = self.lib.aclhip_decompress_tracks_batch(*current["decode_args"])
            if status != 0:
                raise SystemExit(f"the batch launch failed: {status} {self.lib.aclhip_last_error_message(self.context._handle).decode()}")
            current["decoded"].record(self.stream)
            return
        if self._order_args is not None:
            self.order_step()
        status = self._launch(*self._args)
        if status != 0:
            raise SystemExit(f"the batch launch failed: {status} {self.lib.aclhip_last_error_message(self.context._handle).decode()}")

    def caller_update(self, instances, new_clips):
        """the caller's side of an attached list: its own kernel writes the clip changes into its own array, on the decode stream"""
        self.d_clips.index_copy_(0, instances, new_clips)

    def caller_update_ms(self, repeats):
        start, stop = self.torch.cuda.Event(enable_timing=True), self.torch.cuda.Event(enable_timing=True)
        start.record(self.stream)
        for k in range(repeats):
            instances, new_clips, _ = self._updates[k % len(self._updates)]
            self.caller_update(instances, new_clips)
        stop.record(self.stream)
        stop.synchronize()
        return float(start.elapsed_time(stop)) / repeats

    def prewarm(self, seconds):
        """Device pre-warm (setup, not one of the W warm-up steps): an idle MI355X needs a few ms of work before its clocks settle.
        Returns the number of launches it took."""
        launches = 0
        deadline = time.perf_counter() + seconds
        while time.perf_counter() < deadline:
            for _ in range(64):
                self.step()
            launches += 64
            self.torch.cuda.synchronize(self.device)
        return launches

    def kernel_ms(self, repeats):
        """Average device time of one launch: HIP events recorded on the launch stream around `repeats` back-to-back launches."""
        start, stop = self.torch.cuda.Event(enable_timing=True), self.torch.cuda.Event(enable_timing=True)
        start.record(self.stream)
        for _ in range(repeats):
            self.step()
        stop.record(self.stream)
        stop.synchronize()
        return float(start.elapsed_time(stop)) / repeats

    def kernel_name(self):
        if self.is_scalar:
            # 4 instances per wave from 16384 instances on (host_scalar_misc.inl); below that one wave per instance
            return "decompress_scalar_tracks_grouped_kernel" if self.num_instances >= 16384 else "decompress_scalar_tracks_kernel"
        if self.consumers is not None:
            return "decompress_poses_consumer_kernel"
        if self.track_requests:
            return "decompress_track_kernel"          # (ACLHIP_DECODE_FAST changes nothing for single track requests)
        # the library's own answer for this launch (rows of pose_stride bytes, this output descriptor): aclhip_describe_tracks_launch
        return self.context.tracks_kernel_name(self.params, pose_stride_bytes=self.pose_stride, output=self.output)

    def algorithmic_bytes(self):
        """Compulsory-HBM model: bytes written for every instance (in the output layout) + every distinct clip's touched bytes once."""
        written, read = self.context.batch_algorithmic_bytes(self.handles[self.clip_indices])
        if self.track_requests:
            return self.num_instances * (48 + 12) + read              # a 48 byte transform out, clip handle + sample time + track index in, the clip once
        if not self.is_scalar:
            written = written // 48 * self.runtime.LAYOUTS[self.layout][1]
        if self.name == "one_clip_lods":
            written = int(np.minimum(self.track_counts, self.max_tracks).sum()) * self.runtime.LAYOUTS[self.layout][1]       # what the writers keep
        if self.name.startswith("additive_object_space"):
            read += self.context.batch_algorithmic_bytes(self.handles[:1])[1]        # the base clip is read too; one pose per instance is written
        if self.name == "blend_object_space":
            read += self.context.batch_algorithmic_bytes(self.handles[1:])[1]        # the two other clips of every blend; one pose per instance is written
        return written + read

    def stream_schedule(self, steps):
        """database workload: the tiers arrive one chunk at a time, evenly spread over `steps` launches, on the decode stream"""
        if self.database is None:
            return {}
        info = self.context.database_info(self.database)
        runtime = self.runtime
        pending = [(tier, 1) for tier in (runtime.TIER_MEDIUM_IMPORTANCE, runtime.TIER_LOWEST_IMPORTANCE) for _ in range(info.num_chunks[tier - 1])]
        return {(k + 1) * steps // (len(pending) + 1): request for k, request in enumerate(pending)}

    def close(self):
        self.torch.cuda.synchronize(self.device)
        rejected = self.context.rejected_instance_count()
        if self.instance_list is not None:
            self.context.instance_list_destroy(self.instance_list)
        for handle in list(self.handles) + list(self.bystanders):
            self.context.unregister_clip(int(handle))
        if self.database is not None:
            self.context.unregister_database(self.database)
        self.context.close()
        if rejected != 0:
            raise SystemExit(f"the kernel rejected {rejected} instances ({self.name})")


def measured_traffic(key, kernel_name):
    """HBM bytes per launch of the decode kernel from rocprofv3 PMC passes of this same command, committed under profiles/
    (FETCH_SIZE and WRITE_SIZE collected in separate --pmc runs, KiB units, FETCH_SIZE doubled per the gfx950 note of
    MI355X_MICROARCH.md). None when no committed measurement matches the workload and kernel."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    if key is None or not os.path.exists(path):
        return None
    try:
        entries = json.load(open(path))
    except ValueError:
        return None
    for entry in entries:
        if entry.get("workload") == key and entry.get("kernel") == kernel_name:
            return entry.get("traffic_bytes_per_launch")
    return None


# Every Job the default N = 1 run measures outside its timed region, in the order of the line's "workloads" / "layouts" entries:
# (workload, Job options, launches timed). The same list drives the counter passes that measure every entry's HBM traffic (live_traffic).
def default_run_specs():
    return [
        ("one_clip", {}, 300),                                      # the headline batch again (its traffic goes to roofline.traffic)
        ("one_clip_mixed_registry", {}, 300),                       # ... while the context also holds a 300-bone rig and a 551-bone clip
        ("one_clip_lods", {}, 300),                                 # ... with per instance track counts (100 / 60 / 30 bones): poses/s follow the bytes written
        ("256_clips", {}, 300),
        ("256_clips", {"order": "locality"}, 300),
        ("256_clips", {"order": "device"}, 300),                    # ordered on the GPU in front of every launch: the ordering is in kernel_ms
        ("256_clips", {"order": "device_pipelined"}, 300),          # ... the ordering of step k + 1 on a second stream while step k decodes
        ("256_clips", {"order": "list"}, 300),                      # persistent instance list: 1 % of the instances change clip per step, inside the step
        ("256_clips", {"order": "attached"}, 300),                  # ... attached to the caller's clip array: the caller's own kernel writes the 1 %, no update launch
        ("cinematic", {}, 150),
        ("cinematic", {"fast": True}, 150),                         # ACLHIP_DECODE_FAST: rotations within 2e-6, the rest bit identical (opt in)
        ("database", {}, 300),
        ("database", {"order": "locality"}, 300),                   # the same instances laid out in aclhip_order_instances_for_locality order
        ("database", {"order": "list"}, 300),                       # ... kept in a persistent instance list, 1 % changing clip per step
        # single bone requests, scalar track lists, the pose consumers fused into the decode
        ("track_requests", {"num_instances": TRACK_REQUESTS}, 100),
        ("track_requests_256_clips", {"num_instances": TRACK_REQUESTS}, 60),                          # every request its own clip: waves of mixed clips (record heads gathered, round 6)
        ("track_requests_256_clips", {"num_instances": TRACK_REQUESTS, "order": "locality"}, 100),   # ... the same requests in aclhip_order_track_requests_for_locality order: what a caller with a persistent request list does
        ("scalar", {}, 300),
        ("object_space", {}, 150),
        ("additive_object_space", {}, 100),
        ("object_space_fast", {}, 150),                              # the same two with ACLHIP_CONSUMERS_FAST (opt-in arithmetic, <= 2e-6 from the default's poses)
        ("additive_object_space_fast", {}, 100),
        ("blend_object_space", {}, 100),
        ("one_clip", {"layout": "qvv40"}, 300),
        ("one_clip", {"layout": "qv32"}, 300),
    ]


def spec_key(name, options):
    return traffic_key_of(name, options.get("order", "random"), options.get("layout", "qvv48"), fast=options.get("fast", False))


def run_steps(job, steps):
    """`steps` launches of the job; a database workload's tiers arrive chunk by chunk between them, like in the timed regions"""
    schedule = job.stream_schedule(steps)
    for i in range(steps):
        if i in schedule:
            job.context.database_stream_in(job.database, schedule[i][0], schedule[i][1], stream=job.paging_stream.cuda_stream)
        job.step()


def traffic_pass(device_index, manifest_path, steps=12):
    """Child of live_traffic, running under `rocprofv3 --pmc <one counter>`: a few launches of every spec of the default run, a marker
    kernel (the library's plain store sweep over 4 KiB) behind each, and a manifest of what ran (workload key, decode kernel name)."""
    manifest = []
    for name, options, _ in default_run_specs():
        job = Job(name, 0, device_index, **options)
        try:
            run_steps(job, steps)
            job.torch.cuda.synchronize(job.device)
            job.context.measure_write_bandwidth(job.d_poses.data_ptr(), 4096, repeats=1, stream=job.stream.cuda_stream)
            manifest.append({"workload": spec_key(name, options), "kernel": job.kernel_name(), "steps": steps})
        finally:
            job.close()
    json.dump(manifest, open(manifest_path, "w"))


def live_traffic(timeout_s=120):
    """HBM bytes and VALU instructions per launch of the decode kernel of EVERY spec of the default run, measured NOW: this same script
    in three rocprofv3 --pmc passes of their own (traffic_pass above) -- FETCH_SIZE, WRITE_SIZE (KiB; FETCH_SIZE doubled per the gfx950
    note of MI355X_MICROARCH.md) and SQ_INSTS_VALU --, mean over the dispatches of the spec's decode kernel between two markers. Returns
    {workload key: {"traffic": bytes, "valu_instructions": count}} -- empty when rocprofv3 is not there, a profiler is already attached
    or a pass fails (the committed profiles/traffic.json stays)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    rocprof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(rocprof):
        return {}
    # never under a profiler that is already attached to this process (a counter pass next to somebody's trace of the same device)
    if any(name.startswith(("ROCPROF", "ROCP_", "ROCTRACER")) for name in os.environ) or "rocprof" in os.environ.get("LD_PRELOAD", ""):
        return {}
    means = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE", "SQ_INSTS_VALU"):
        directory = tempfile.mkdtemp(prefix="aclhip_pmc_", dir="/tmp")
        try:
            manifest_path = os.path.join(directory, "manifest.json")
            command = [rocprof, "--pmc", counter, "--output-format", "csv", "-d", directory, "-o", "pass", "--", sys.executable, os.path.abspath(__file__), "--traffic-pass", manifest_path]
            environment = dict(os.environ, ACLHIP_BENCH_PROFILING="1", TMPDIR="/tmp")
            subprocess.run(command, cwd="/tmp", env=environment, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=timeout_s, check=True)
            manifest = json.load(open(manifest_path))
            rows = []
            for path in glob.glob(os.path.join(directory, "**", "*counter_collection.csv"), recursive=True):
                rows += [(int(row["Dispatch_Id"]), row["Kernel_Name"], float(row["Counter_Value"])) for row in csv.DictReader(open(path)) if row["Counter_Name"] == counter]
            rows.sort()
            groups, current, in_marker = [], [], False
            for _, kernel, value in rows:
                if "stream_write_kernel" in kernel:
                    if not in_marker:
                        groups.append(current)
                        current = []
                    in_marker = True
                else:
                    in_marker = False
                    current.append((kernel, value))
            if len(groups) != len(manifest):
                raise ValueError("the counter pass saw other launches than its manifest lists")
            for entry, group in zip(manifest, groups):
                values = [value for kernel, value in group if entry["kernel"] in kernel]
                if values:
                    means.setdefault(entry["workload"], {})[counter] = sum(values) / len(values)
        except (OSError, subprocess.SubprocessError, KeyError, ValueError):
            if counter != "SQ_INSTS_VALU":       # (the HBM passes are what "traffic" needs; the instruction count is an extra)
                return {}
        finally:
            shutil.rmtree(directory, ignore_errors=True)
    # SQ_INSTS_VALU: vector ALU instructions the decode kernel issues per launch (all waves) -- the VALU issue floor of the entry
    return {key: {"traffic": int((2.0 * value["FETCH_SIZE"] + value["WRITE_SIZE"]) * 1024.0), "valu_instructions": value.get("SQ_INSTS_VALU")}
            for key, value in means.items() if "FETCH_SIZE" in value and "WRITE_SIZE" in value}


def traffic_key_of(workload, order, layout, keep_rows=False, fast=False):
    if keep_rows:
        return None
    key = workload
    if order != "random":
        key += f", {order} order"
    if layout != "qvv48":
        key += f", {layout}"
    if fast:
        key += ", fast"
    return key


def measure_job(name, rank, device_index, repeats=300, **job_options):
    """One entry of "workloads" / "footprint_sweep" / "layouts": the kernel's mean launch time (HIP events on the launch stream)
    over `repeats` launches after a short pre-warm, against the algorithmic bytes of the batch."""
    job = Job(name, rank, device_index, **job_options)
    try:
        job.prewarm(0.05)
        if job.database is not None:
            # the low importance tier arrives chunk by chunk between the launches, like in the headline database run
            start, stop = job.torch.cuda.Event(enable_timing=True), job.torch.cuda.Event(enable_timing=True)
            start.record(job.stream)
            run_steps(job, repeats)
            stop.record(job.stream)
            stop.synchronize()
            kernel_ms = float(start.elapsed_time(stop)) / repeats
        else:
            kernel_ms = job.kernel_ms(repeats)
        ordering_ms_device = decode_ms_order_reused = None
        if job.order == "device":
            start, stop = job.torch.cuda.Event(enable_timing=True), job.torch.cuda.Event(enable_timing=True)
            start.record(job.stream)
            for _ in range(repeats):
                job.order_step()
            stop.record(job.stream)
            stop.synchronize()
            ordering_ms_device = float(start.elapsed_time(stop)) / repeats
            order_arguments, job._order_args = job._order_args, None          # the instance list outlives the frame: ordered once, decoded again and again
            decode_ms_order_reused = job.kernel_ms(repeats)
            job._order_args = order_arguments
        algorithmic = job.algorithmic_bytes()
        achieved = algorithmic / (kernel_ms * 1e-3) / 1e9
        kernel = job.kernel_name()
        achievable = None
        if not job.is_scalar and not job.track_requests and job.layout == "qvv48":
            achievable = job.context.measure_pose_store_bandwidth(job.d_poses.data_ptr(), job.pose_stride, job.num_instances, job.max_tracks, repeats=10, stream=job.stream.cuda_stream)[0]
        return {
            "workload": traffic_key_of(name, job.order, job.layout, job.keep_rows, job.fast) or f"{name}, {job.order} order, rows kept",
            "config": WORKLOAD_TEXT[name],
            "order": job.order,
            "layout": job.layout,
            "instances": job.num_instances,
            "bones": job.max_tracks,
            "distinct_clips": len(job.clips),
            "pose_bytes": job.pose_stride,
            "kernel": kernel,
            "kernel_ms": kernel_ms,
            "poses_per_s": job.num_instances / (kernel_ms * 1e-3),
            "algorithmic_bytes": int(algorithmic),
            "achieved": achieved,
            "frac": achieved / HBM_PEAK_GBPS,
            "best_store_only_gbps": achievable,        # this batch's own write stream alone, best of thirteen shapes (aclhip_measure_pose_store_bandwidth): a second denominator, not a bound
            "frac_of_best_store_only": None if not achievable else achieved / achievable,
            "traffic": measured_traffic(traffic_key_of(name, job.order, job.layout, job.keep_rows, job.fast), kernel),
            "traffic_source": "profiles/traffic.json (committed rocprofv3 --pmc passes)",
            "ordering_ms_host": None if job.ordering_ms is None else round(job.ordering_ms, 3),
            "ordering_ms_device": ordering_ms_device,          # inside kernel_ms when the order is "device"
            "kernel_ms_order_reused": decode_ms_order_reused,  # the decode alone in that order (an instance list ordered once, sample times refreshed per frame)
            **bound_of(kernel_ms, int(algorithmic), None),      # (the instruction floor arrives with the live counter passes of the default run)
            "caller_update_ms": job.caller_update_ms(repeats) if job.order == "attached" else None,      # order "attached": the caller's kernel alone (inside kernel_ms)
            "list_orderings": None if job.instance_list is None else int(job.context.instance_list_order(job.instance_list)[1]),     # order "list": times the library (re-)ordered the list, setup included
            "registration_ms_total": round(job.registration_ms, 3),      # validate + derive tables + upload for all of the workload's clips (setup)
            "launches_timed": repeats,
        }
    finally:
        job.close()


NUM_SIMDS = 256 * 4             # MI355X: 256 CUs x 4 SIMDs (MI355X_MICROARCH.md)
PEAK_CLOCK_HZ = 2.4e9           # the peak engine clock: a floor computed with it is a floor at any clock the device really ran at


def bound_of(kernel_ms, algorithmic_bytes, valu_instructions):
    """Which resource binds an entry, and how close to THAT bound it runs. Two floors under a launch: its algorithmic bytes at the HBM
    specification rate, and its vector ALU instructions (SQ_INSTS_VALU of this run's counter pass) issued at one wave64 instruction per
    four cycles per SIMD on all 1 024 SIMDs at the peak clock -- optimistic on purpose (no transcendental quarter rates, no dependent
    issue stalls, every SIMD busy from the first to the last cycle), so that frac_of_bound never flatters. The HBM fraction of a kernel
    whose instruction floor is the higher one says little about the kernel; this field says which one to read."""
    hbm_floor_ms = algorithmic_bytes / (HBM_PEAK_GBPS * 1e9) * 1e3
    valu_floor_ms = None if not valu_instructions else valu_instructions * 4.0 / (NUM_SIMDS * PEAK_CLOCK_HZ) * 1e3
    bound = "valu" if valu_floor_ms is not None and valu_floor_ms > hbm_floor_ms else "hbm"
    floor = max(hbm_floor_ms, valu_floor_ms or 0.0)
    return {"bound": bound, "valu_instructions": None if not valu_instructions else int(valu_instructions), "valu_issue_floor_ms": valu_floor_ms, "hbm_floor_ms": hbm_floor_ms,
            "frac_of_bound": floor / kernel_ms}


def measure_database_paging(rank, device_index, repeats=300):
    """The database workload with its tiers paged in on a SECOND stream while the batches run (BASELINE.json configs[4]: "paged from host
    DRAM via pinned hipMemcpyAsync"; the reference lets a stream-in run next to decompression, database_streamer.impl.h:60-136), beside
    the same schedule issued on the decode stream and the decode alone in the two end states. Every figure is the decode stream's own
    time per launch (HIP events on it). The pose buffer after the paged run must equal the one of a decode with everything resident."""
    job = Job("database", rank, device_index, paging="second_stream")
    torch, context, database = job.torch, job.context, job.database
    try:
        def timed(paging_stream):
            start, stop = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            job.paging_stream = paging_stream
            torch.cuda.synchronize(job.device)
            start.record(job.stream)
            run_steps(job, repeats)
            stop.record(job.stream)
            torch.cuda.synchronize(job.device)
            return float(start.elapsed_time(stop)) / repeats

        def everything_out():
            for tier in (job.runtime.TIER_MEDIUM_IMPORTANCE, job.runtime.TIER_LOWEST_IMPORTANCE):
                context.database_stream_out(database, tier, stream=job.stream.cuda_stream)
            torch.cuda.synchronize(job.device)

        second_stream = job.paging_stream
        job.prewarm(0.05)
        # (a stream's first use creates its hardware queue, milliseconds: the second stream pages everything in and out once before anything is timed)
        for tier in (job.runtime.TIER_MEDIUM_IMPORTANCE, job.runtime.TIER_LOWEST_IMPORTANCE):
            context.database_stream_in(database, tier, stream=second_stream.cuda_stream)
        second_stream.synchronize()
        everything_out()
        nothing_streamed_ms = job.kernel_ms(repeats)
        same_stream_ms = timed(job.stream)
        everything_out()
        second_stream_ms = timed(second_stream)
        paged_poses = job.d_poses.clone()
        everything_streamed_ms = job.kernel_ms(repeats)
        torch.cuda.synchronize(job.device)
        same_poses = bool(torch.equal(paged_poses.view(torch.int32), job.d_poses.view(torch.int32)))
        # the copies alone: everything out, then in again back to back on the second stream
        everything_out()
        info = context.database_info(database)
        start, stop = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        start.record(second_stream)
        for tier in (job.runtime.TIER_MEDIUM_IMPORTANCE, job.runtime.TIER_LOWEST_IMPORTANCE):
            context.database_stream_in(database, tier, stream=second_stream.cuda_stream)
        stop.record(second_stream)
        stop.synchronize()
        copy_ms = float(start.elapsed_time(stop))
        paged_bytes = int(sum(info.bulk_data_size))
        algorithmic = job.algorithmic_bytes()
        return {
            "workload": "database, paged on a second stream",
            "config": WORKLOAD_TEXT["database"].replace("on the decode stream", "on a second stream"),
            "instances": job.num_instances, "kernel": job.kernel_name(), "launches_timed": repeats,
            "kernel_ms": second_stream_ms,                                  # decode stream time per launch while the tiers arrive on the other stream
            "kernel_ms_paged_on_the_decode_stream": same_stream_ms,         # round 4's protocol: every copy between two launches
            "kernel_ms_nothing_streamed": nothing_streamed_ms,              # the decode alone, before any tier has arrived ...
            "kernel_ms_everything_streamed": everything_streamed_ms,       # ... and after all of them have
            "paging_overhead": second_stream_ms / (0.5 * (nothing_streamed_ms + everything_streamed_ms)) - 1.0,     # against the mean of the two end states the paged run passes through
            "poses_equal_everything_resident": same_poses,
            "paged_bytes": paged_bytes, "paged_chunks": int(sum(info.num_chunks)), "h2d_ms_back_to_back": copy_ms,
            "h2d_gbps": None if copy_ms <= 0 else paged_bytes / (copy_ms * 1e-3) / 1e9,
            "algorithmic_bytes": int(algorithmic), "achieved": algorithmic / (second_stream_ms * 1e-3) / 1e9, "frac": algorithmic / (second_stream_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
            "traffic": None, "traffic_source": None,
        }
    finally:
        job.paging_stream = job.stream
        job.close()


def self_check(job, count=256):
    """After the timed region: `count` random rows of the pose buffer the LAST timed launch wrote, against the CPU oracle for the same
    (clip, sample time) -- the driver's number certifies itself (the reference's harness validates before it times,
    tools/acl_compressor/sources/validate_tracks.cpp:92-260). Pose batches in a caller-visible row order, single track requests and scalar
    lists; None for what has no oracle on this side (database tiers, consumers) or no fixed rows (device / list orders)."""
    from oracle import bindings as ob      # the checker, never the thing measured
    if job.database is not None or job.consumers is not None or job.order in ("device", "device_pipelined", "list", "attached") or job.d_rows is not None or job.name == "one_clip_lods" or job.fast:
        return None
    torch = job.torch
    torch.cuda.synchronize(job.device)
    rng = np.random.default_rng(12345)
    rows = np.sort(rng.choice(job.num_instances, size=min(count, job.num_instances), replace=False))
    got = job.d_poses[torch.from_numpy(rows).to(job.device)].cpu().numpy()
    blobs = [c.blob for c in job.clips]
    which, times = job.clip_indices[rows].astype(np.uint32), job.times[rows]
    if job.is_scalar:
        expected = ob.oracle_scalar_decompress_tracks_batch(blobs, which, times, got.shape[1])
    elif job.track_requests:
        tracks = job.d_tracks.cpu().numpy()[rows]
        expected = np.stack([ob.oracle_decompress_track(blobs[int(c)], float(t), int(b)) for c, t, b in zip(which, times, tracks)])
    else:
        full = ob.oracle_decompress_tracks_batch(blobs, which, times, job.max_tracks)
        layout_id, bytes_per_track = job.runtime.LAYOUTS[job.layout]
        expected = job.runtime.relayout_pose(full, layout_id).reshape(rows.size, -1)
        got = got[:, : job.max_tracks * bytes_per_track // 4]
    got = got.reshape(expected.shape)
    finite = np.isfinite(expected) & np.isfinite(got)
    return {"instances": int(rows.size), "max_abs_err": float(np.abs(got[finite] - expected[finite]).max()) if finite.any() else 0.0,
            "bit_exact": bool(np.array_equal(got.view(np.uint32), expected.view(np.uint32))),
            "against": "oracle/acl_oracle.c (held to the reference's own headers bit for bit by tests/test_oracle_vs_reference.py), rows of the last timed launch"}


# ---- CPU baseline -----------------------------------------------------------------------------------------------------------

def cgroup_cpu_max():
    """The container's CPU quota: "max" or "<quota> <period>" of cgroup v2 (cpu.max), or the v1 pair; None when unreadable."""
    try:
        return open("/sys/fs/cgroup/cpu.max").read().strip()
    except OSError:
        pass
    try:
        quota = open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read().strip()
        period = open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read().strip()
        return f"{quota} {period}"
    except OSError:
        return None


def cpu_topology():
    """(sockets, cores per socket, threads per core) of the host from /proc/cpuinfo and sysfs -- what lscpu prints; Nones when unreadable"""
    try:
        physical, cores = set(), {}
        package = None
        for line in open("/proc/cpuinfo"):
            key, _, value = line.partition(":")
            key, value = key.strip(), value.strip()
            if key == "physical id":
                package = int(value)
                physical.add(package)
            elif key == "cpu cores" and package is not None:
                cores[package] = int(value)
        sockets = len(physical)
        cores_per_socket = max(cores.values()) if cores else None
        siblings = open("/sys/devices/system/cpu/cpu0/topology/thread_siblings_list").read().strip()
        threads_per_core = len(siblings.replace("-", ",").split(",")) if "," in siblings or "-" in siblings else 1
        if "-" in siblings and "," not in siblings:
            low, high = siblings.split("-")
            threads_per_core = int(high) - int(low) + 1
        return (sockets or None), cores_per_socket, threads_per_core
    except (OSError, ValueError):
        return None, None, None


def cpu_baseline(clips, clip_indices, times, row_units, is_scalar):
    """Times the reference's own decoder (oracle/_ref, kind "reference") -- or the C restatement (kind "port") when the
    reference build is absent -- on the same instance list on this box's host cores: PINNED threads that start together and decode
    for a fixed wall-clock window per point (oracle/bench_harness.h), swept over thread counts; `value` is the best point."""
    from oracle import bindings as ob  # cpu_baseline leg: the oracle is the baseline being measured here, never the product

    nproc = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    sample = min(clip_indices.size, 65536)
    indices = np.ascontiguousarray(clip_indices[:sample], dtype=np.uint32)
    sample_times = np.ascontiguousarray(times[:sample], dtype=np.float32)
    blob_ptrs = (ctypes.c_void_p * len(clips))(*[c.blob.ctypes.data for c in clips])
    description = {"unit": "poses/s", "nproc": nproc, "cgroup_cpu_max": cgroup_cpu_max()}

    have_reference = ob.have_ref_scalar() if is_scalar else ob.have_ref()
    if have_reference:
        lib = ob.ref_scalar() if is_scalar else ob.ref()
        bench = lib.aclref_scalar_bench_timed if is_scalar else lib.aclref_bench_timed
        seconds = float(os.environ.get("ACLHIP_BENCH_CPU_SECONDS", "1.0"))
        points = sorted({t for t in (1, 8, 32, 64, 128, 256) if t < nproc} | {nproc})
        sweep = {}
        for threads in points:
            # the single-thread point carries the extrapolation below: best of three windows (round 4's driver run had ONE one-second
            # window land below the cold-cache figure -- a neighbour on the box, not physics)
            windows = 3 if threads == points[0] else 1
            sweep[str(threads)] = max(bench(blob_ptrs, indices.ctypes.data, sample_times.ctypes.data, sample, row_units, threads, seconds, 1, None) for _ in range(windows))
        best_threads = max(sweep, key=lambda k: sweep[k])
        # the reference's OWN benchmark protocol next to the warm sweep (tools/acl_decompressor/sources/benchmark.cpp:232-281): one thread, 220
        # copies of the clip decoded in turn, the CPU caches flushed between sample times -- what its published numbers are measured with
        cold = None
        if not is_scalar and hasattr(lib, "aclref_bench_cold"):
            blob = clips[int(indices[0])].blob
            cold_times = np.ascontiguousarray(sample_times[:100], dtype=np.float32)
            cold = lib.aclref_bench_cold(blob.ctypes.data, blob.size, cold_times.ctypes.data, cold_times.size, row_units, 220, 128 << 20, seconds)
        per_thread = sweep[str(points[0])] / points[0]
        sockets, cores_per_socket, threads_per_core = cpu_topology()
        physical_cores = sockets * cores_per_socket if sockets and cores_per_socket else None
        # one thread at its best: the warm loop or the reference's own cold-cache protocol, whichever is faster on this box
        best_1t = max(per_thread, cold or 0.0)
        description.update({
            "cold_cache_1t": cold,                               # poses/s of ONE thread under the reference's cold-cache protocol (first clip of the list)
            "sockets": sockets, "cores_per_socket": cores_per_socket, "threads_per_core": threads_per_core, "physical_cores": physical_cores,       # lscpu style, beside `nproc` logical CPUs
            # what this host would reach if every one of its `nproc` logical CPUs ran at the best single-thread rate measured (an upper bound: SMT
            # siblings share a core, memory bandwidth is ignored) -- next to `value`, which is what the container's CPU quota really gives;
            # the same per PHYSICAL core is the fairer figure for an SMT host
            "extrapolated_all_cpus": best_1t * nproc,
            "extrapolated_physical_cores": None if physical_cores is None else best_1t * physical_cores,
            "extrapolated_from": "max(per_thread_1t: best of three windows, cold_cache_1t)",
            "value": sweep[best_threads], "cores": int(best_threads), "threads_at_best": int(best_threads), "kind": "reference",
            "per_thread_1t": per_thread, "sweep": sweep,
            "sample": f"{sample} instances of the same list statically partitioned over the threads, seek + decompress_tracks, reference headers "
                      f"({'default_scalar_decompression_settings' if is_scalar else 'AVX2 build, benchmark settings (benchmark.cpp:94-101)'}), one context and one private output "
                      f"buffer per thread (warm cache), threads pinned to the CPUs of the affinity mask and started together, {seconds:g} s wall clock per point of the thread sweep",
        })
        return description

    lib = ob.oracle()
    options = ob.default_options()
    sample = min(sample, 4096 if is_scalar else 16384)
    best = 1e30
    if is_scalar:
        out = np.zeros(row_units, dtype=np.float32)
        t0 = time.perf_counter()
        for i in range(sample):
            lib.aclo_scalar_decompress_tracks(clips[indices[i]].blob.ctypes.data, ctypes.c_float(sample_times[i]), 0, ctypes.byref(options), out.ctypes.data)
        best = time.perf_counter() - t0
    else:
        out = np.zeros((sample, row_units * 12), dtype=np.float32)
        for _ in range(3):
            t0 = time.perf_counter()
            lib.aclo_decompress_tracks_batch(blob_ptrs, indices.ctypes.data, sample_times.ctypes.data, sample, 0, ctypes.byref(options), out.ctypes.data, row_units * 12)
            best = min(best, time.perf_counter() - t0)
    description.update({"value": sample / best, "cores": 1, "threads_at_best": 1, "kind": "port", "per_thread_1t": sample / best, "sweep": {"1": sample / best},
                        "sample": f"{sample} instances of the same list, scalar C restatement (oracle/acl_oracle.c), 1 thread"})
    return description


# ---- pose gather (N > 1) ----------------------------------------------------------------------------------------------------

def measure_gather(job, dist, rank, world_size, mode, repeats=5):
    """Times the gather of every rank's pose shard, separately from the decode: "rccl" = one all-gather (every rank receives every
    shard, torch.distributed = RCCL over xGMI); "p2p" = every rank copies its shard into rank 0's buffer through a peer mapping
    (hipIpc + hipMemcpyAsync device to device: rank 0's seven inbound links work concurrently, nothing else moves).
    Returns the entries of the "gather" object; errors are reported, never raised (the decode numbers stand on their own)."""
    from acl_amd import sharding
    torch = job.torch
    shard = job.d_poses
    shard_bytes = shard.numel() * 4
    on_device = dist.get_backend() == "nccl"
    result = {"shard_bytes": int(shard_bytes)}
    modes = ("rccl", "p2p") if mode == "both" else (mode,)

    def timed(enqueue):
        torch.cuda.synchronize(job.device)
        dist.barrier()
        best = 1e30
        for _ in range(repeats):
            dist.barrier()
            t0 = time.perf_counter()
            enqueue()
            torch.cuda.synchronize(job.device)
            dist.barrier()
            best = min(best, time.perf_counter() - t0)
        slowest = torch.tensor([best], dtype=torch.float64, device=job.device if on_device else "cpu")
        dist.all_reduce(slowest, op=dist.ReduceOp.MAX)
        return float(slowest.item()) * 1e3

    if "rccl" in modes:
        try:
            source = shard if on_device else shard.cpu()
            gathered = torch.empty((world_size * shard.shape[0], shard.shape[1]), dtype=shard.dtype, device=source.device)
            ms = timed(lambda: dist.all_gather_into_tensor(gathered, source))
            result["rccl_all_gather_ms"] = ms
            # every rank receives (W - 1) shards
            result["rccl_all_gather_gbps_into_each_rank"] = (world_size - 1) * shard_bytes / (ms * 1e-3) / 1e9
            del gathered
        except Exception as error:      # noqa: BLE001 -- reported in the line
            result["rccl_all_gather_error"] = repr(error)[:300]
    if "p2p" in modes:
        gather = None
        try:
            gather = sharding.PeerGather(job.context, shard_bytes, rank, world_size, device=job.device)
            ms = timed(lambda: gather.push(shard.data_ptr(), stream=job.stream.cuda_stream))
            result["p2p_to_rank0_ms"] = ms
            result["p2p_gbps_into_rank0"] = (world_size - 1) * shard_bytes / (ms * 1e-3) / 1e9     # rank 0's own shard is a local copy
            result["p2p_gbps_per_link"] = shard_bytes / (ms * 1e-3) / 1e9                          # each remote shard travels over its own link
            result["p2p_link_frac_of_153_gbps"] = result["p2p_gbps_per_link"] / XGMI_LINK_GBPS
        except Exception as error:      # noqa: BLE001
            result["p2p_error"] = repr(error)[:300]
        finally:
            if gather is not None:
                gather.close()
    return result


def measure_sharded_job(name, rank, world_size, device_index, dist, gather_mode, repeats, num_instances=INSTANCES_PER_GPU):
    """N > 1: one of the 8-GPU configs of BASELINE.json (configs[3]: the 300-bone rig; configs[4]: database-bound clips whose tiers are
    streamed in) on every rank's shard, timed like the headline -- barrier, `repeats` launches, device synchronized, barrier, MAX over
    the ranks -- then the gathers of the shards. The database's residency advances on all ranks together
    (acl_amd.sharding.stream_database_everywhere: rank 0's request is broadcast) inside the timed loop."""
    from acl_amd import sharding
    job = Job(name, rank, device_index, num_instances=num_instances)
    torch = job.torch
    on_device = dist.get_backend() == "nccl"
    try:
        job.prewarm(0.05)
        schedule = job.stream_schedule(repeats)
        start, stop = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(job.device)
        dist.barrier()
        t0 = time.perf_counter()
        start.record(job.stream)
        for i in range(repeats):
            if i in schedule:
                sharding.stream_database_everywhere(job.context, job.database, schedule[i][0], schedule[i][1], stream=job.stream.cuda_stream)
            job.step()
        stop.record(job.stream)
        torch.cuda.synchronize(job.device)
        dist.barrier()
        elapsed = time.perf_counter() - t0
        times = torch.tensor([elapsed, float(start.elapsed_time(stop)) * 1e-3], dtype=torch.float64, device=job.device if on_device else "cpu")
        dist.all_reduce(times, op=dist.ReduceOp.MAX)
        elapsed, kernel_seconds = (float(v) for v in times.tolist())
        kernel_ms = kernel_seconds * 1e3 / repeats
        algorithmic = job.algorithmic_bytes()
        entry = {
            "workload": name, "config": WORKLOAD_TEXT[name], "n_gpus": world_size, "instances_per_gpu": job.num_instances, "bones": job.max_tracks,
            "distinct_clips": len(job.clips), "pose_bytes": job.pose_stride, "kernel": job.kernel_name(),
            "launches_timed": repeats, "ms_per_step": elapsed / repeats * 1e3,
            "poses_per_s": job.num_instances * world_size * repeats / elapsed,          # whole job: every rank's shard over the slowest rank's time
            "kernel_ms": kernel_ms,                                                     # slowest rank, HIP events on its launch stream
            "algorithmic_bytes_per_gpu": int(algorithmic), "achieved_per_gpu": algorithmic / (kernel_ms * 1e-3) / 1e9,
            "frac": algorithmic / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
            "database_chunks_streamed_in_together": len(schedule),
        }
        if gather_mode != "none":
            entry["gather"] = measure_gather(job, dist, rank, world_size, gather_mode)
        return entry
    finally:
        job.close()


def device_pci_bus_id(torch, device_index):
    """PCI address of the device (what hipDeviceGetPCIBusId prints), from the runtime torch already holds; None when it does not tell.
    (No second copy of the HIP runtime is opened for this: two runtimes in one process do not see the same devices.)"""
    properties = torch.cuda.get_device_properties(device_index)
    if hasattr(properties, "pci_bus_id"):
        return f"{getattr(properties, 'pci_domain_id', 0):04x}:{properties.pci_bus_id:02x}:{getattr(properties, 'pci_device_id', 0):02x}"
    return getattr(properties, "uuid", None) and str(properties.uuid)


def distributed_checks(torch, dist, rank, world_size, device_index, backend, kernel_ms):
    """N > 1, every rank: what the first run on a real 8-GPU node should prove about itself before its numbers are read -- the
    communicator has N ranks that can reduce (one all-reduce of ones on the device), the ranks sit on N DISTINCT GPUs (PCI bus ids),
    every pair of GPUs this process sees can map each other's memory (what the peer gather needs), and no rank is a straggler
    (kernel time of the timed region per rank). Returns the "checks" object of the line; nothing here raises."""
    import socket
    on_device = backend == "nccl"
    checks = {"backend": backend, "n_gpus_claimed": world_size, "ok": True, "problems": []}
    try:
        ones = torch.ones(1, dtype=torch.float32, device=torch.device("cuda", device_index) if on_device else "cpu")
        dist.all_reduce(ones)
        checks["communicator_ranks"] = int(round(float(ones.item())))         # RCCL's own count of the ranks that took part
        identities = [None] * world_size
        dist.all_gather_object(identities, (socket.gethostname(), device_pci_bus_id(torch, device_index), int(device_index)))
        checks["devices"] = [f"{host}/{bus}" for host, bus, _ in identities]
        checks["distinct_devices"] = len({(host, bus) for host, bus, _ in identities})
        visible = torch.cuda.device_count()
        peers = [bool(torch.cuda.can_device_access_peer(device_index, other)) for other in range(visible) if other != device_index]
        all_peers = [None] * world_size
        dist.all_gather_object(all_peers, (int(sum(peers)), len(peers)))
        checks["peer_access"] = [f"{have}/{of}" for have, of in all_peers]
        per_rank = [None] * world_size
        dist.all_gather_object(per_rank, float(kernel_ms))
        checks["kernel_ms_per_rank"] = per_rank
        checks["kernel_ms_min"], checks["kernel_ms_max"] = min(per_rank), max(per_rank)
        if checks["communicator_ranks"] != world_size:
            checks["problems"].append(f"the all-reduce counted {checks['communicator_ranks']} ranks, not {world_size}")
        if on_device and checks["distinct_devices"] != world_size:
            checks["problems"].append(f"{world_size} ranks on {checks['distinct_devices']} distinct GPUs")
        if on_device and any(have != of for have, of in all_peers):
            checks["problems"].append("some pair of GPUs cannot map each other's memory: the peer gather is skipped")
        if checks["kernel_ms_max"] > 1.15 * checks["kernel_ms_min"]:
            checks["problems"].append(f"straggler: rank {per_rank.index(max(per_rank))} takes {checks['kernel_ms_max'] / checks['kernel_ms_min']:.2f} x the fastest rank's kernel time")
    except Exception as error:      # noqa: BLE001 -- reported in the line; the decode numbers stand on their own
        checks["problems"].append(repr(error)[:300])
    checks["ok"] = not checks["problems"]
    return checks



# ---- the line the driver reads --------------------------------------------------------------------------------------------------

HEADLINE_LIMIT = 4096          # bytes: the LAST stdout line stays below this (round 5's 26 KB line was not parsed by the driver)
DETAILS_PATH = os.environ.get("ACLHIP_BENCH_DETAILS", os.path.join(ROOT, "bench_details.json"))      # (tests point it at their tmp_path)
WORKLOAD_COLUMNS = ["kernel_ms", "frac", "bound", "frac_of_bound", "traffic_ratio"]


def _sig(value, digits=6):
    """floats at `digits` significant digits (the line is read by people and by a size-limited parser); everything else unchanged"""
    if isinstance(value, bool) or not isinstance(value, float):
        return value
    if value != value or value in (float("inf"), float("-inf")):
        return None
    return float(f"{value:.{digits}g}")


def _pick(source, keys):
    return {key: _sig(source[key]) for key in keys if source is not None and key in source}


def workload_row(entry):
    """one entry of "workloads" / "layouts" / "footprint_sweep" as [kernel_ms, frac, bound, frac_of_bound, traffic / algorithmic bytes]"""
    algorithmic = entry.get("algorithmic_bytes") or entry.get("algorithmic_bytes_per_gpu")
    traffic = entry.get("traffic")
    return [_sig(entry.get("kernel_ms"), 4), _sig(entry.get("frac"), 3), entry.get("bound"), _sig(entry.get("frac_of_bound"), 3),
            None if not traffic or not algorithmic else _sig(traffic / algorithmic, 4)]


def compact_headline(result):
    """The ONE line the driver parses, from the full result: the contract's keys, `roofline`, `cpu_baseline`, `self_check` and one row per
    extra workload -- everything else (per-entry configurations, sweeps, sources, the thread sweep of the CPU baseline, the gather's
    breakdown) lives in bench_details.json next to this file and on stderr. Always below HEADLINE_LIMIT bytes: rows are dropped from the
    end of the workload map (and said so) before the limit is crossed."""
    line = _pick(result, ["metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data"])
    config = result.get("config") or {}
    line["config"] = _pick(config, ["workload", "instances_per_gpu", "bones", "distinct_clips", "layout", "pose_bytes"])
    if "workload" in line["config"]:
        line["config"]["workload"] = line["config"]["workload"][:240]
    line["config"]["parallelism"] = f"instances sharded over {result.get('n_gpus', 1)} GPU(s), no collective on the data path"
    roofline = result.get("roofline") or {}
    line["roofline"] = _pick(roofline, ["bound", "achieved", "peak", "unit", "frac", "traffic", "algorithmic_bytes_per_launch", "kernel", "kernel_ms",
                                        "valu_issue_floor_ms", "hbm_floor_ms", "frac_of_bound", "best_store_only_gbps"])
    if roofline.get("kernel_ms_per_rank"):
        line["roofline"]["kernel_ms_per_rank"] = [_sig(v, 4) for v in roofline["kernel_ms_per_rank"]]
    cpu = result.get("cpu_baseline")
    if cpu is not None:
        line["cpu_baseline"] = _pick(cpu, ["value", "unit", "cores", "threads_at_best", "nproc", "physical_cores", "cgroup_cpu_max", "per_thread_1t", "cold_cache_1t", "kind",
                                           "gpu_over_cpu", "gpu_over_cpu_extrapolated_physical_cores"])
        line["cpu_baseline"]["sample"] = (cpu.get("sample") or "")[:200]
    check = result.get("self_check")
    line["self_check"] = check if check is None or "error" in check else _pick(check, ["instances", "max_abs_err", "bit_exact"])
    if result.get("checks") is not None:
        checks = result["checks"]
        line["checks"] = _pick(checks, ["backend", "ok", "communicator_ranks", "distinct_devices", "kernel_ms_min", "kernel_ms_max"])
        line["checks"]["problems"] = [str(problem)[:120] for problem in checks.get("problems", [])[:4]]
    if result.get("gather") is not None:
        line["gather"] = {key: (_sig(value, 5) if not isinstance(value, str) else value[:120]) for key, value in result["gather"].items()
                          if isinstance(value, (int, float, str))}
    rows = {}
    for entry in result.get("layouts") or []:
        if entry.get("layout") and entry["layout"] != "qvv48":
            rows[f"one_clip, {entry['layout']}"] = workload_row(entry)
    for entry in result.get("workloads") or []:
        if "error" in entry:
            rows[str(entry.get("workload"))] = [None, None, "error", None, None]
        else:
            rows[str(entry.get("workload"))] = workload_row(entry)
    for entry in result.get("footprint_sweep") or []:
        rows[f"one_clip, {entry['instances']} instances"] = workload_row(entry)
    line["workload_columns"] = WORKLOAD_COLUMNS
    line["workloads"] = rows
    line["details"] = "bench_details.json (next to bench.py; the same object is printed on stderr)"
    dropped = 0
    while len(json.dumps(line)) >= HEADLINE_LIMIT - 64 and line["workloads"]:
        line["workloads"].pop(next(reversed(line["workloads"])))
        dropped += 1
    if dropped:
        line["workloads_dropped_for_size"] = dropped
    if len(json.dumps(line)) >= HEADLINE_LIMIT:       # (cannot happen with the keys above; the contract's keys survive whatever does)
        line["config"]["workload"] = line["config"].get("workload", "")[:80]
        line.pop("gather", None)
        line.pop("checks", None)
    return line


def emit(result):
    """rank 0: the full result to bench_details.json and stderr, then the compact headline as the LAST (and only) line on stdout"""
    try:
        with open(DETAILS_PATH, "w") as out:
            json.dump(result, out)
            out.write("\n")
    except OSError:
        pass
    print(json.dumps(result), file=sys.stderr, flush=True)
    print(json.dumps(compact_headline(result)), flush=True)


def relaunch_under_torchrun(gpus):
    """`python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment: this process becomes
    `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port <free> bench.py <same arguments>`
    (one rank per GPU over RCCL), which is how the driver launches N > 1 itself."""
    import socket
    with socket.socket() as probe:
        probe.bind(("127.0.0.1", 0))
        port = probe.getsockname()[1]
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    command = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={gpus}", "--master-addr", "127.0.0.1", "--master-port", str(port),
               os.path.abspath(__file__)] + sys.argv[1:]
    sys.stdout.flush()
    sys.stderr.flush()
    os.execv(sys.executable, command)


def main():
    parser = argparse.ArgumentParser()
    parser.add_argument("--gpus", type=int, default=1)
    # the clocks of an idle MI355X take a few ms to ramp: the defaults warm up for ~30 ms and time ~0.1 s
    parser.add_argument("--steps", type=int, default=2000)
    parser.add_argument("--warmup", type=int, default=500)
    parser.add_argument("--workload", default="one_clip", choices=sorted(WORKLOAD_TEXT))
    parser.add_argument("--instances", type=int, default=INSTANCES_PER_GPU, help="instances per GPU")
    parser.add_argument("--order", default="random", choices=["random", "by_clip", "locality", "device", "device_pipelined", "list", "attached"],
                        help="instance order: as drawn; bucketed by clip on the host; aclhip_order_instances_for_locality (host, setup); "
                             "aclhip_order_instances_device in front of every launch (part of the step)")
    parser.add_argument("--no-live-traffic", action="store_true", help="default run: keep roofline.traffic from profiles/traffic.json instead of measuring it with two rocprofv3 --pmc passes")
    parser.add_argument("--keep-rows", action="store_true", help="with --order locality: store every pose in its instance's ORIGINAL row")
    parser.add_argument("--layout", default="qvv48", choices=["qvv48", "qvv40", "qv32"], help="output layout (aclhip_output_desc)")
    parser.add_argument("--fast", action="store_true", help="ACLHIP_DECODE_FAST: the opt in tolerance mode (rotations within 2e-6 of the bit exact kernels')")
    parser.add_argument("--no-cpu-baseline", action="store_true", help="skip the cpu_baseline leg")
    parser.add_argument("--no-extras", action="store_true", help="only the headline workload: no other workloads, footprint sweep, layouts")
    parser.add_argument("--gather", default="both", choices=["none", "rccl", "p2p", "both"], help="N > 1: time the pose gather after the decode (reported separately)")
    parser.add_argument("--traffic-pass", default=None, metavar="MANIFEST", help="internal (live_traffic): a few launches of every workload of the default run under rocprofv3 --pmc")
    # kept for the scripts of round 1
    parser.add_argument("--sort-by-clip", action="store_true", help="same as --order by_clip")
    parser.add_argument("--order-for-locality", action="store_true", help="same as --order locality")
    args = parser.parse_args()
    if args.sort_by_clip:
        args.order = "by_clip"
    if args.order_for_locality:
        args.order = "locality"

    import torch
    import torch.distributed as dist

    world_size = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    distributed = world_size > 1
    if args.gpus != world_size and distributed:
        raise SystemExit(f"--gpus {args.gpus} does not match WORLD_SIZE {world_size}")
    if args.gpus > 1 and not distributed:
        # (ACLHIP_BENCH_BACKEND=gloo dry runs share GPUs between ranks; the real thing needs one GPU per rank)
        if os.environ.get("ACLHIP_BENCH_BACKEND", "nccl") == "nccl" and torch.cuda.is_available() and torch.cuda.device_count() < args.gpus:
            raise SystemExit(f"--gpus {args.gpus} on a box with {torch.cuda.device_count()} GPU(s)")
        relaunch_under_torchrun(args.gpus)

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

    profiling = os.environ.get("ACLHIP_BENCH_PROFILING", "0") == "1"
    if args.traffic_pass is not None:
        traffic_pass(device_index, args.traffic_pass)
        return
    instances = TRACK_REQUESTS if args.workload in ("track_requests", "track_requests_256_clips") and args.instances == INSTANCES_PER_GPU else args.instances
    job = Job(args.workload, rank, device_index, num_instances=instances, order=args.order, keep_rows=args.keep_rows, layout=args.layout, fast=args.fast)

    # device pre-warm (skipped under a counter-collecting profiler, where every launch is serialized and slow: ACLHIP_BENCH_PROFILING=1)
    prewarm_launches = 0 if profiling else job.prewarm(0.15)
    for _ in range(args.warmup):
        job.step()

    # two HIP events on the launch stream bracket the timed launches: their distance / K is the kernel's mean duration including the
    # (~1 us) launch boundary (no events in between: every record is one more packet between two launches)
    start_mark, stop_mark = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    stream_schedule = job.stream_schedule(args.steps)

    if distributed:
        dist.barrier()
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    start_mark.record(job.stream)
    for i in range(args.steps):
        if i in stream_schedule:
            job.context.database_stream_in(job.database, stream_schedule[i][0], stream_schedule[i][1], stream=job.paging_stream.cuda_stream)
        job.step()
    stop_mark.record(job.stream)
    # The closing bracket: the device synchronized, then the barrier. The clock stops when THIS rank's K steps are done -- the MAX over
    # the ranks below is the job's time -- and not behind the barrier: a barrier is an all-reduce of its own (tens of microseconds over
    # RCCL), the data path has no collective, and the driver's K = 20 steps are a millisecond. The time with the barrier inside is in
    # the line as well (ms_per_step_behind_closing_barrier: two ranks over gloo on one GPU, K = 20: 0.1142 against 0.1047 ms).
    # (Polling the stop event instead of the blocking synchronize changes nothing: 50.8 - 52.3 us per step at K = 20 either way.)
    torch.cuda.synchronize(device)
    elapsed = time.perf_counter() - t0
    if distributed:
        dist.barrier()
    elapsed_behind_barrier = time.perf_counter() - t0

    if distributed:
        elapsed_tensor = torch.tensor([elapsed, elapsed_behind_barrier], dtype=torch.float64, device=device if backend == "nccl" else "cpu")
        dist.all_reduce(elapsed_tensor, op=dist.ReduceOp.MAX)
        elapsed, elapsed_behind_barrier = (float(v) for v in elapsed_tensor.tolist())

    # the rows the LAST TIMED launch wrote, against the oracle -- before anything else touches the pose buffer, outside every timed region
    checked = None
    if rank == 0 and not profiling:
        try:
            checked = self_check(job)
        except Exception as error:      # noqa: BLE001 -- reported in the line, the timing stands on its own
            checked = {"error": repr(error)[:300]}

    # Roofline of the decode kernel: device time per launch from the HIP events of the timed region
    kernel_ms = float(start_mark.elapsed_time(stop_mark)) / args.steps
    checks = distributed_checks(torch, dist, rank, world_size, device_index, backend, kernel_ms) if distributed else None
    if not profiling:
        job.prewarm(0.03)       # (the self check read poses back: the device idled for a moment)
    kernel_ms_back_to_back = None if profiling else job.kernel_ms(max(10, min(args.steps, 100)))
    algorithmic_bytes = job.algorithmic_bytes()
    achieved_gbps = algorithmic_bytes / (kernel_ms * 1e-3) / 1e9
    fill_gbps = job.context.measure_write_bandwidth(job.d_poses.data_ptr(), job.num_instances * job.pose_stride, repeats=1 if profiling else 20, stream=job.stream.cuda_stream)
    # the ceiling of THIS batch's write stream on THIS device: the pose kernels' own store pattern, nothing decoded, best occupancy
    achievable_gbps = achievable_waves = None
    if not job.is_scalar and not job.track_requests and job.layout == "qvv48":
        achievable_gbps, achievable_waves = job.context.measure_pose_store_bandwidth(job.d_poses.data_ptr(), job.pose_stride, job.num_instances, job.max_tracks,
                                                                                     repeats=1 if profiling else 20, stream=job.stream.cuda_stream)

    kernel_name = job.kernel_name()
    result = None
    if rank == 0:
        total_poses = job.num_instances * world_size * args.steps
        workload_text = WORKLOAD_TEXT[args.workload]
        if args.order != "random":
            workload_text += {"by_clip": ", bucketed by clip", "locality": ", decoded in aclhip_order_instances_for_locality order",
                              "device": ", ordered by aclhip_order_instances_device in front of every launch (inside the step)",
                              "device_pipelined": ", ordered by aclhip_order_instances_device on a second stream while the previous step decodes (inside the step)",
                              "list": ", kept in a persistent aclhip_instance_list: 1 % of the instances change clip in every step (inside the step), the library re-orders when 1/8 has changed",
                              "attached": ", kept in an aclhip_instance_list ATTACHED to the caller's clip array: every step a caller side kernel (torch index_copy_) writes 1 % of the clips there (inside the step), no update launch, the library re-orders when 1/8 has changed"}[args.order]
            workload_text += ", poses scattered back to their original rows" if args.keep_rows else ""
        result = {
            "metric": "poses/sec (whole node), 64k clip instances x 100 bones per GPU, seek + decompress_tracks",
            "value": total_poses / elapsed,
            "unit": "poses/s",
            "n_gpus": world_size,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "ms_per_step_behind_closing_barrier": elapsed_behind_barrier / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": workload_text,
                "instances_per_gpu": int(job.num_instances),
                "bones": int(job.max_tracks),
                "distinct_clips": len(job.clips),
                "layout": args.layout,
                "ordering_ms": None if job.ordering_ms is None else round(job.ordering_ms, 3),     # aclhip_order_instances_for_locality on the host (setup, not timed)
                "registration_ms_total": round(job.registration_ms, 3),      # validate + derive tables + upload, all clips (setup, not timed)
                "prewarm_launches": int(prewarm_launches),                   # untimed launches before the W warm-up steps (clock ramp)
                "pose_bytes": int(job.pose_stride),
                "sharding": f"instances split over {world_size} rank(s), no collective on the data path",
            },
            "roofline": {
                "bound": "hbm",
                "achieved": achieved_gbps,
                "peak": HBM_PEAK_GBPS,
                "unit": "GB/s",
                "frac": achieved_gbps / HBM_PEAK_GBPS,
                "traffic": measured_traffic(traffic_key_of(args.workload, args.order, args.layout, args.keep_rows), kernel_name),
                "kernel": kernel_name,
                "kernel_ms": kernel_ms,
                "kernel_ms_back_to_back": kernel_ms_back_to_back,
                "algorithmic_bytes_per_launch": int(algorithmic_bytes),
                "plain_store_stream_gbps": fill_gbps,      # a plain 16 B per lane store sweep over the same pose buffer, for scale (not a ceiling: DESIGN.md 6)
                # the batch's own write stream alone (one wave per pose window, the kernels' 1 KiB streaming stores, nothing decoded): the best
                # of thirteen store-only shapes (four occupancies x three pacings, the runtime's fill). A second denominator next to the
                # specification's 8 TB/s, NOT a bound: a decode whose stores are paced by its own loads can come out above it
                "best_store_only_gbps": achievable_gbps,
                "best_store_only_waves_per_cu": achievable_waves,
                "frac_of_best_store_only": None if not achievable_gbps else achieved_gbps / achievable_gbps,
            },
        }
    if rank == 0:
        result["self_check"] = checked
    if rank == 0 and checks is not None:
        result["checks"] = checks
        result["roofline"]["kernel_ms_per_rank"] = checks.get("kernel_ms_per_rank")
    if distributed and args.gather != "none" and checks is not None and any("peer gather is skipped" in problem for problem in checks["problems"]):
        args.gather = "rccl" if args.gather in ("both", "rccl") else "none"
    if distributed and args.gather != "none":
        # The gathers are timed AFTER the decode numbers are final, under a watchdog: the decode line must reach the driver whatever a
        # collective or a peer mapping does on a node this code has never run on. A gather that does not come back within the limit is
        # reported as such and the line is printed from the watchdog thread (the main thread may be stuck inside the runtime).
        import threading
        gather = {"status": "running"}
        if rank == 0:
            result["gather"] = gather
        finished = threading.Event()

        def watchdog():
            if finished.wait(float(os.environ.get("ACLHIP_BENCH_GATHER_TIMEOUT", "120"))):
                return
            if rank == 0:
                gather["status"] = "timed out"
                emit(result)
            os._exit(0)

        threading.Thread(target=watchdog, daemon=True).start()
        gather.update(measure_gather(job, dist, rank, world_size, args.gather))
        gather["status"] = "done"

    headline = (job.clips, job.clip_indices, job.times, job.pose_stride // 4 if job.is_scalar else job.max_tracks, job.is_scalar, job.database is not None, job.consumers is not None)
    job.close()

    if distributed and not args.no_extras and not profiling and args.workload == "one_clip":
        # BASELINE.json's 8-GPU configs on this many GPUs: the 300-bone rig (65 536 instances per GPU, 944 MB shards: the gathers that
        # matter) and the database-bound clips (32 768 per GPU, tiers streamed in on all ranks together). Still under the watchdog.
        sharded = []
        if rank == 0:
            result["workloads"] = sharded
        # (ACLHIP_BENCH_SHARDED_INSTANCES: instances per GPU of the rig shard, the database shard takes half; dry runs on one GPU use it)
        sharded_instances = int(os.environ.get("ACLHIP_BENCH_SHARDED_INSTANCES", str(INSTANCES_PER_GPU)))
        for name, instances, repeats in (("cinematic", sharded_instances, 100), ("database", max(sharded_instances // 2, 1), 200)):
            try:
                sharded.append(measure_sharded_job(name, rank, world_size, device_index, dist, args.gather, repeats, num_instances=instances))
            except Exception as error:      # noqa: BLE001 -- reported in the line; the headline stands on its own
                sharded.append({"workload": name, "error": repr(error)[:300]})
    if distributed and args.gather != "none":
        finished.set()

    extras = world_size == 1 and not args.no_extras and not profiling and args.workload == "one_clip" and args.order == "random" and args.layout == "qvv48" and args.instances == INSTANCES_PER_GPU and not args.fast
    if rank == 0 and extras:
        # the other north-star configs, measured in this process outside the timed region (about 0.2 s of launches each)
        specs = default_run_specs()
        entries = [measure_job(name, rank, device_index, repeats=repeats, **options) for name, options, repeats in specs]
        paging_entry = measure_database_paging(rank, device_index)
        result["roofline"]["traffic_source"] = "profiles/traffic.json (committed rocprofv3 --pmc passes)"
        if not args.no_live_traffic:
            # EVERY entry's HBM traffic measured by THIS run (two counter passes of a few launches of every spec, after the timed regions)
            # instead of read back from profiles/traffic.json
            measured = live_traffic()
            source = "this run: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE passes of 12 launches of every workload, FETCH_SIZE x 2 (gfx950)"
            for entry in entries:
                if entry["workload"] in measured:
                    entry["traffic_committed"], entry["traffic"], entry["traffic_source"] = entry["traffic"], measured[entry["workload"]]["traffic"], source
                    entry.update(bound_of(entry["kernel_ms"], entry["algorithmic_bytes"], measured[entry["workload"]]["valu_instructions"]))
            if "one_clip" in measured:
                result["roofline"]["traffic_committed"], result["roofline"]["traffic"], result["roofline"]["traffic_source"] = result["roofline"]["traffic"], measured["one_clip"]["traffic"], source
                floors = bound_of(kernel_ms, int(algorithmic_bytes), measured["one_clip"]["valu_instructions"])
                result["roofline"].update({key: floors[key] for key in ("valu_instructions", "valu_issue_floor_ms", "hbm_floor_ms", "frac_of_bound")})
                result["roofline"]["bound"] = floors["bound"]
        layout_entries = [entries[0]] + [e for e in entries if e["workload"] in ("one_clip, qvv40", "one_clip, qv32")]
        result["workloads"] = [e for e in entries[1:] if e not in layout_entries] + [paging_entry]
        result["layouts"] = [{key: entry[key] for key in ("layout", "pose_bytes", "kernel", "kernel_ms", "poses_per_s", "achieved", "frac", "algorithmic_bytes", "traffic", "traffic_source")} for entry in layout_entries]
        result["footprint_sweep"] = [
            {key: entry[key] for key in ("instances", "kernel_ms", "poses_per_s", "achieved", "frac", "algorithmic_bytes")}
            for entry in (measure_job("one_clip", rank, device_index, num_instances=n, repeats=200) for n in (32768, 65536, 131072))]

    if rank == 0:
        if world_size == 1 and not args.no_cpu_baseline:
            clips, clip_indices, times, row_units, is_scalar, has_database, has_consumers = headline
            result["cpu_baseline"] = cpu_baseline(clips, clip_indices, times, row_units, is_scalar)
            if has_database:
                # the reference's database_context is not part of the CPU bridge that travels to the GPU box
                result["cpu_baseline"]["sample"] += "; clips bound WITHOUT their database (highest importance tier only)"
            if has_consumers:
                result["cpu_baseline"]["sample"] += "; decode of the (additive) clip only, the consumers are not part of the CPU timing"
            result["cpu_baseline"]["gpu_over_cpu"] = result["value"] / result["cpu_baseline"]["value"]
            if result["cpu_baseline"].get("extrapolated_all_cpus"):
                result["cpu_baseline"]["gpu_over_cpu_extrapolated"] = result["value"] / result["cpu_baseline"]["extrapolated_all_cpus"]
            if result["cpu_baseline"].get("extrapolated_physical_cores"):
                result["cpu_baseline"]["gpu_over_cpu_extrapolated_physical_cores"] = result["value"] / result["cpu_baseline"]["extrapolated_physical_cores"]
        emit(result)

    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
