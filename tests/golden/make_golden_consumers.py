#!/usr/bin/env python3
"""Generates tests/golden/consumers/*.npz: what the REFERENCE's own pose consumers make of poses decoded by the REFERENCE's own
decoder -- acl::apply_additive_to_base (core/additive_utils.h:150) and acl::local_to_object_space
(compression/transform_pose_utils.h:35), built from /root/reference by oracle/Makefile (oracle/_ref/libaclref_pose.so) -- for
clips produced by the reference's compressor from synthetic raw animation.

Run in the build container (where /root/reference exists):  python tests/golden/make_golden_consumers.py
Each case stores: additive clip blob, base clip blob, parent indices, (additive time, base time) pairs, and per additive format
(none / relative / additive0 / additive1) the combined local poses and their object space poses [n, num_tracks, 12].
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from acl_amd import synth  # noqa: E402
from oracle import bindings as ob  # noqa: E402

OUT_DIR = os.path.join(HERE, "consumers")

# name: (num_tracks, num_samples of the additive clip / of the base clip, has_scale, how far back a parent may be, extra roots)
CASES = {
    "biped_40_scale": dict(seed=61, num_tracks=40, samples=(45, 31), has_scale=1, parent_span=6, extra_roots=0),
    "rig_100_two_roots": dict(seed=62, num_tracks=100, samples=(64, 20), has_scale=0, parent_span=10, extra_roots=1),
    "crowd_rig_400_multi_window": dict(seed=63, num_tracks=400, samples=(24, 24), has_scale=1, parent_span=40, extra_roots=3),
    "chain_12": dict(seed=64, num_tracks=12, samples=(10, 10), has_scale=1, parent_span=1, extra_roots=0),
}


def make_hierarchy(rng, num_tracks, parent_span, extra_roots):
    parents = np.zeros(num_tracks, dtype=np.uint32)
    parents[0] = ob.INVALID_PARENT
    for i in range(1, num_tracks):
        parents[i] = rng.integers(max(0, i - parent_span), i)
    for i in rng.choice(np.arange(1, num_tracks), size=extra_roots, replace=False):
        parents[i] = ob.INVALID_PARENT
    return parents


def make_raw(rng, num_tracks, num_samples, has_scale, additive):
    """A smooth raw animation [num_samples, num_tracks, 12]; additive clips hold small deltas around identity / zero / one"""
    t = np.linspace(0.0, 1.0, num_samples, dtype=np.float32)[:, None, None]
    phase = rng.uniform(0, 2 * np.pi, size=(1, num_tracks, 3)).astype(np.float32)
    speed = rng.uniform(0.5, 3.0, size=(1, num_tracks, 3)).astype(np.float32)
    wave = np.sin(phase + speed * t * 2 * np.pi).astype(np.float32)
    raw = np.zeros((num_samples, num_tracks, 12), dtype=np.float32)
    amplitude = 0.15 if additive else 0.6
    xyz = wave * amplitude
    w = np.sqrt(np.maximum(1.0 - (xyz * xyz).sum(axis=2, keepdims=True), 0.0))
    raw[:, :, 0:3] = xyz
    raw[:, :, 3:4] = w
    offset = 0.0 if additive else rng.uniform(-0.4, 0.4, size=(1, num_tracks, 3))
    raw[:, :, 4:7] = offset + np.roll(wave, 1, axis=2) * (0.05 if additive else 0.2)
    if has_scale:
        raw[:, :, 8:11] = 1.0 + np.roll(wave, 2, axis=2) * (0.05 if additive else 0.1)
    else:
        raw[:, :, 8:11] = 1.0
    return raw.astype(np.float32)


def main():
    if not (ob.have_ref() and ob.have_ref_compressor() and ob.have_ref_pose()):
        raise SystemExit("oracle/_ref is incomplete: run `make -C oracle ref` where /root/reference exists")
    os.makedirs(OUT_DIR, exist_ok=True)
    for name, case in CASES.items():
        rng = np.random.default_rng(case["seed"])
        num_tracks = case["num_tracks"]
        parents = make_hierarchy(rng, num_tracks, case["parent_span"], case["extra_roots"])
        compressor_parents = np.where(parents == ob.INVALID_PARENT, -1, parents).astype(np.int32)
        additive_blob = ob.ref_compress(make_raw(rng, num_tracks, case["samples"][0], case["has_scale"], True), 30.0, parents=compressor_parents)
        base_blob = ob.ref_compress(make_raw(rng, num_tracks, case["samples"][1], case["has_scale"], False), 30.0, parents=compressor_parents)

        n = 12 if num_tracks < 400 else 5
        durations = [ob.ref().aclref_get_duration(blob.ctypes.data, -1) for blob in (additive_blob, base_blob)]
        times = np.stack([np.concatenate([rng.uniform(0.0, d, size=n - 2), [0.0, d]]) for d in durations], axis=1).astype(np.float32)

        local = np.zeros((4, n, num_tracks, 12), dtype=np.float32)
        object_space = np.zeros_like(local)
        for i in range(n):
            additive_pose = np.zeros((num_tracks, 12), dtype=np.float32)
            base_pose = np.zeros((num_tracks, 12), dtype=np.float32)
            ob.ref_decompress(additive_blob, float(times[i, 0]), 0, -1, 0, 0, -1, None, None, out=additive_pose)
            ob.ref_decompress(base_blob, float(times[i, 1]), 0, -1, 0, 0, -1, None, None, out=base_pose)
            for pose in (additive_pose, base_pose):     # W lanes of translation / scale are unspecified in the reference
                pose[:, 7] = 0.0
                pose[:, 11] = 0.0
            for additive_format in range(4):
                combined = ob.ref_apply_additive_to_base(additive_format, base_pose, additive_pose)
                combined[:, 7] = 0.0
                combined[:, 11] = 0.0
                local[additive_format, i] = combined
                # the reference's local_to_object_space knows one root (transform 0): walk every root's subtree through it by
                # giving further roots an identity parent -- same arithmetic for their descendants, the roots themselves pass through
                object_space[additive_format, i] = reference_object_space(parents, combined)

        path = os.path.join(OUT_DIR, f"{name}.npz")
        np.savez_compressed(path, additive_blob=np.asarray(additive_blob), base_blob=np.asarray(base_blob), parents=parents, times=times,
                            local=local, object_space=object_space)
        print(f"{name}: {os.path.getsize(path)} bytes, {num_tracks} tracks, additive {additive_blob.size} + base {base_blob.size} bytes")


def reference_object_space(parents, local_pose):
    """acl::local_to_object_space per root: transforms are renumbered so that each root's subtree is a pose of its own"""
    num_tracks = parents.size
    root_of = np.zeros(num_tracks, dtype=np.int64)
    for i in range(num_tracks):
        root_of[i] = i if (i == 0 or parents[i] == ob.INVALID_PARENT) else root_of[parents[i]]
    out = np.zeros_like(local_pose)
    for root in np.unique(root_of):
        members = np.flatnonzero(root_of == root)           # ascending: still parent first
        renumber = {int(m): k for k, m in enumerate(members)}
        sub_parents = np.array([0] + [renumber[int(parents[m])] for m in members[1:]], dtype=np.uint32)
        result = ob.ref_local_to_object_space(sub_parents, local_pose[members])
        result[:, 7] = 0.0
        result[:, 11] = 0.0
        out[members] = result
    return out


if __name__ == "__main__":
    main()
