#!/usr/bin/env python3
"""Measurement / confidence aid (GPU box): random clip shapes x random settings through the C ABI against the CPU oracle, for a given
number of seconds. Poses (both kernel families, every rounding / looping policy, normalization, per instance rounding), single bone
requests, the compact layouts, object space and additive consumers with random hierarchies. Prints the first mismatch and exits 1.

usage: python tools/fuzz_gpu.py [seconds] [seed]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from acl_amd import runtime, synth          # noqa: E402
from oracle import bindings as ob           # noqa: E402


def same(got, expected):
    """bit identical, except that a NaN may answer a NaN (x86's default NaN is negative, the GPU's positive: only poses that overflowed
    -- additive scales multiplied down a 100 level chain -- get there)"""
    a, b = got.view(np.uint32), expected.view(np.uint32)
    return bool(np.all((a == b) | (np.isnan(got) & np.isnan(expected))))


def random_spec(rng):
    tracks = int(rng.choice([1, 2, 7, 33, 64, 100, 104, 105, 130, 209, 320, 700]))
    samples = int(rng.choice([1, 2, 3, 16, 17, 40, 120, 400]))
    spec = dict(seed=int(rng.integers(1, 1 << 30)), num_tracks=tracks, num_samples=samples, sample_rate=float(rng.choice([24.0, 30.0, 60.0, 29.97])),
                rotation_default=float(rng.uniform(0, 0.3)), rotation_constant=float(rng.uniform(0, 0.6)),
                translation_default=float(rng.uniform(0, 0.5)), translation_constant=float(rng.uniform(0, 0.5)),
                raw_fraction=float(rng.choice([0.0, 0.01, 0.2])), width0_fraction=float(rng.choice([0.0, 0.03, 0.3])),
                wrap=int(rng.integers(0, 2)), strip_keyframes=int(rng.integers(0, 2)) if samples > 3 else 0,
                min_bits=int(rng.integers(1, 10)), max_bits=int(rng.integers(10, 20)))
    if rng.uniform() < 0.5:
        spec.update(has_scale=1, scale_default=float(rng.uniform(0, 0.8)), scale_constant=float(rng.uniform(0, 0.2)))
        if rng.uniform() < 0.2:
            spec.update(mirrored_scale_fraction=0.3)
    return spec


def main():
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    deadline = time.time() + seconds
    rounds = checks = 0
    os.environ["ACLHIP_FORCE_GENERIC_KERNEL"] = "1"
    generic = runtime.Context(0)            # pinned to the any-settings kernels
    os.environ.pop("ACLHIP_FORCE_GENERIC_KERNEL")
    with runtime.Context(0) as common, generic:
        while time.time() < deadline:
            context = generic if rng.integers(0, 2) else common
            spec = random_spec(rng)
            clip = synth.build_clip(**spec)
            other = synth.build_clip(**dict(spec, seed=spec["seed"] + 1, num_samples=max(1, spec["num_samples"] // 2 + 1)))
            clips = [clip, other]
            handles = np.array([context.register_clip(c.blob) for c in clips], dtype=np.uint32)
            tracks = spec["num_tracks"]
            n = int(rng.integers(1, 40))
            which = rng.integers(0, 2, size=n)
            times = np.array([rng.uniform(-0.1, clips[w].duration + 0.1) for w in which], dtype=np.float32)
            rounding = int(rng.integers(0, 4))
            looping = int(rng.integers(0, 3))
            normalization = int(rng.integers(0, 3))
            params = runtime.default_params(rounding_policy=rounding, looping_policy=looping, normalization=normalization)
            options = ob.default_options(looping_policy=looping, normalization=normalization)
            poses = context.decompress_tracks(handles[which], times, params=params)
            for i in range(n):
                expected = ob.oracle_decompress_tracks(clips[which[i]].blob, float(times[i]), rounding, options)
                if not same(poses[i], expected):
                    print("POSE MISMATCH", spec, "instance", i, "time", times[i], "rounding", rounding, "looping", looping, "normalization", normalization)
                    return 1
            checks += n
            # every instance with a looping policy and a writer of its own (ABI 5): its first K tracks through one of three skip masks
            if rng.uniform() < 0.5:
                import ctypes
                instance_looping = rng.integers(0, 3, size=n).astype(np.uint8)
                counts = rng.integers(0, tracks + 2, size=n).astype(np.uint32)
                mask_table = rng.integers(0, 8, size=(3, tracks)).astype(np.uint8)
                mask_table[0] = 0
                instance_masks = rng.integers(0, 3, size=n).astype(np.uint8)
                writer_params = runtime.default_params(rounding_policy=rounding, looping_policy=looping, normalization=normalization)
                writer_params.instance_looping_policies = instance_looping.ctypes.data
                output = runtime.OutputDesc()
                output.mask_table, output.mask_stride, output.instance_masks, output.instance_track_counts = mask_table.ctypes.data, tracks, instance_masks.ctypes.data, counts.ctypes.data
                fill = np.float32(7.0)
                got = np.full((n, tracks, 12), fill, dtype=np.float32)
                instance_handles = np.ascontiguousarray(handles[which])
                context._check(context._lib.aclhip_decompress_tracks_host_out(context._handle, instance_handles.ctypes.data, times.ctypes.data, n, ctypes.byref(writer_params), 0,
                                                                              ctypes.byref(output), got.ctypes.data, tracks * 48))
                for i in range(n):
                    expected = np.full((tracks, 12), fill, dtype=np.float32)
                    full = ob.oracle_decompress_tracks(clips[which[i]].blob, float(times[i]), rounding, ob.default_options(looping_policy=int(instance_looping[i]), normalization=normalization))
                    kept = min(int(counts[i]), tracks)
                    expected[:kept] = full[:kept]
                    for kind in range(3):
                        expected[((mask_table[instance_masks[i]] >> kind) & 1) == 1, kind * 4: kind * 4 + 4] = fill
                    if not same(got[i], expected):
                        print("WRITER MISMATCH", spec, "instance", i, "time", times[i], "count", int(counts[i]), "mask", int(instance_masks[i]), "looping", int(instance_looping[i]), "rounding", rounding, "normalization", normalization)
                        return 1
                checks += n
            # every instance with its own table of per track rounding policies (track_writer::get_rounding_policy per pose), poses and single tracks
            if tracks > 0 and rng.uniform() < 0.3:
                tables = rng.integers(0, 4, size=(3, tracks)).astype(np.uint8)
                table_of = rng.integers(0, 3, size=n).astype(np.uint8)
                table_params = runtime.default_params(rounding_policy=runtime.ROUND_PER_TRACK, per_track_rounding=1, looping_policy=looping, normalization=normalization)
                table_params.track_rounding_table, table_params.track_rounding_stride, table_params.instance_rounding_tables = tables.ctypes.data, tracks, table_of.ctypes.data
                got = context.decompress_tracks(handles[which], times, params=table_params)
                wanted = rng.integers(0, tracks, size=n)
                single = context.decompress_track(handles[which], times, wanted, params=table_params)
                for i in range(n):
                    table_options = ob.default_options(looping_policy=looping, normalization=normalization, per_track_rounding=1)
                    row = np.ascontiguousarray(tables[table_of[i]])
                    table_options.track_rounding = row.ctypes.data
                    expected = ob.oracle_decompress_tracks(clips[which[i]].blob, float(times[i]), ob.ROUND_PER_TRACK, table_options)
                    expected_single = ob.oracle_decompress_track(clips[which[i]].blob, float(times[i]), int(wanted[i]), rounding=ob.ROUND_PER_TRACK, options=table_options)
                    if not same(got[i], expected) or not same(single[i], expected_single):
                        print("ROUNDING TABLE MISMATCH", spec, "instance", i, "time", times[i], "table", int(table_of[i]), "track", int(wanted[i]), "looping", looping, "normalization", normalization)
                        return 1
                checks += 2 * n
            # object space + additive, when the consumers take these settings
            if normalization != 2 and tracks <= 700:
                parents = np.zeros(tracks, dtype=np.uint32)
                parents[0] = runtime.NO_PARENT
                for t in range(1, tracks):
                    parents[t] = rng.integers(max(0, t - 9), t)
                for handle in handles:
                    context.set_clip_hierarchy(int(handle), parents)
                additive_format = int(rng.integers(0, 4))
                base_which = rng.integers(0, 2, size=n)
                base_times = np.array([rng.uniform(0.0, clips[w].duration) for w in base_which], dtype=np.float32)
                kwargs = dict(additive_format=additive_format, params=params, object_space=bool(rng.integers(0, 2)))
                if additive_format != 0:
                    kwargs.update(base_clips=handles[base_which], base_sample_times=base_times)
                got = context.decompress_poses(handles[which], times, **kwargs)
                for i in range(n):
                    local = ob.oracle_decompress_tracks(clips[which[i]].blob, float(times[i]), rounding, options)
                    if additive_format != 0:
                        base = ob.oracle_decompress_tracks(clips[base_which[i]].blob, float(base_times[i]), rounding, options)
                        local = ob.oracle_apply_additive_to_base(additive_format, base, local)
                    expected = ob.oracle_local_to_object_space(parents, local) if kwargs["object_space"] else local
                    if not same(got[i], expected):
                        print("CONSUMER MISMATCH", spec, "instance", i, kwargs["object_space"], additive_format, "rounding", rounding, "looping", looping, "normalization", normalization)
                        return 1
                checks += n
            # single bone requests (decompress_track): the whole pose's bits
            if tracks > 0:
                wanted = rng.integers(0, tracks, size=n)
                single = context.decompress_track(handles[which], times, wanted, params=params)
                for i in range(n):
                    expected = ob.oracle_decompress_tracks(clips[which[i]].blob, float(times[i]), rounding, options)[wanted[i]]
                    if not same(single[i], expected):
                        print("TRACK MISMATCH", spec, "instance", i, "track", int(wanted[i]), "rounding", rounding, "looping", looping, "normalization", normalization)
                        return 1
                checks += n
            # a blend of 2 .. 4 clip instances (both clips have the same tracks)
            if normalization != 2 and tracks <= 700 and rng.uniform() < 0.5:
                k = int(rng.integers(2, 5))
                others = rng.integers(0, 2, size=(n, k - 1))
                other_times = np.array([[rng.uniform(0.0, clips[w].duration) for w in row] for row in others], dtype=np.float32)
                weights = rng.dirichlet(np.ones(k), size=n).astype(np.float32)
                object_space = bool(rng.integers(0, 2))
                got = context.decompress_poses(handles[which], times, params=params, object_space=object_space,
                                               blend_clips=handles[others], blend_sample_times=other_times, blend_weights=weights)
                expected = ob.oracle_decompress_blended_poses_batch([c.blob for c in clips], which, times, others, other_times, weights, tracks,
                                                                    parent_indices=parents if object_space else None, rounding=rounding, options=options)
                for i in range(n):
                    if not same(got[i], expected[i]):
                        print("BLEND MISMATCH", spec, "instance", i, "k", k, object_space, "rounding", rounding, "looping", looping, "normalization", normalization)
                        return 1
                checks += n
            for handle in handles:
                context.unregister_clip(int(handle))
            # a scalar track list now and then
            if rng.uniform() < 0.3:
                scalar_spec = dict(seed=int(rng.integers(1, 1 << 30)), track_type=int(rng.integers(0, 5)), num_tracks=int(rng.choice([1, 3, 64, 65, 256, 300, 1000])),
                                   num_samples=int(rng.choice([1, 2, 30, 200])), raw_fraction=float(rng.choice([0.0, 0.05, 0.3])), wrap=int(rng.integers(0, 2)))
                curves = synth.build_scalar_clip(**scalar_spec)
                handle = context.register_clip(curves.blob)
                count = int(rng.integers(1, 30))
                curve_times = rng.uniform(-0.1, curves.duration + 0.1, size=count).astype(np.float32)
                values = context.decompress_scalar_tracks(np.full(count, handle, dtype=np.uint32), curve_times, params=params)
                for i in range(count):
                    expected = ob.oracle_scalar_decompress_tracks(curves.blob, float(curve_times[i]), rounding, options)
                    if not same(values[i, : curves.num_tracks], expected):
                        print("SCALAR MISMATCH", scalar_spec, "instance", i, "time", curve_times[i], "rounding", rounding, "looping", looping)
                        return 1
                checks += count
                context.unregister_clip(handle)
            rounds += 1
        print(f"fuzz ok: {rounds} clip pairs, {checks} poses checked, rejected {common.rejected_instance_count() + generic.rejected_instance_count()}")
    return 0


if __name__ == "__main__":
    sys.exit(main())
