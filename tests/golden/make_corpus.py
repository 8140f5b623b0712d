#!/usr/bin/env python3
"""Generates tests/golden/corpus/*.npz: a corpus of compressed_tracks written by the REFERENCE's own compressor
(oracle/_ref/libaclref_compress.so = compress_track_list, oracle/_ref/libaclref_db.so = build_database; built from /root/reference by
oracle/Makefile) from synthetic raw animation, spanning what the reference's regression set spans
(/root/reference/test_data/configs/*.config.sjson x its clips, tools/acl_compressor/sources/validate_tracks.cpp:92-260):

  * the 13 regression configurations: compression level medium / high / highest, the raw and the two mixed format configurations
    (quatf_full, vector3f_full next to the variable formats), matrix error metric, bind pose relative clips, keyframe stripping,
    and the four database configurations (default 1 MiB / 4 KiB chunks, medium 0.3 + low 0.4 tiers);
  * loop optimisation, scale / no scale, mirrored (negative) scale, 1 .. 600 samples (1, 2, 3, 16, 17, 31, 32, 33 ... around the
    segmenting limits), 1 .. 551 bones (Trooper_Main has 551), several sample rates and precisions, optional metadata;
  * motion that stresses the decoder's arithmetic: hinge joints (1e-7 noise on idle axes), rotations near half a turn (W near 0),
    fast spins, translations of +-1000 units, constant and default (identity / bind pose) sub-tracks in any proportion.

Only DATA is stored: compressed blobs (plus the skeleton and bind pose they were compressed with, for the metadata tests) and the
specification each came from. No expected poses: the tests decode every sample x every bone with the C oracle, which
tests/test_corpus_oracle.py holds to the reference's own decoder bit for bit over this same corpus wherever oracle/_ref exists.
v02_00_00 blobs cannot come from here (the compressor writes the latest version only): those stay synthetic (tests/golden/v2_0_low_bits.npz).

Run in the build container (where /root/reference exists):  python tests/golden/make_corpus.py
"""
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import bindings as ob  # noqa: E402

OUT = os.path.join(HERE, "corpus")
NO_PARENT = -1


# ---- raw animation ----------------------------------------------------------------------------------------------------------

def quat_mul(a, b):
    """Hamilton product of [..., 4] xyzw quaternions"""
    ax, ay, az, aw = np.moveaxis(a, -1, 0)
    bx, by, bz, bw = np.moveaxis(b, -1, 0)
    return np.stack([aw * bx + ax * bw + ay * bz - az * by,
                     aw * by - ax * bz + ay * bw + az * bx,
                     aw * bz + ax * by - ay * bx + az * bw,
                     aw * bw - ax * bx - ay * by - az * bz], axis=-1)


def quat_from_axis_angle(axis, angle):
    axis = axis / np.maximum(np.linalg.norm(axis, axis=-1, keepdims=True), 1e-20)
    half = 0.5 * angle[..., None]
    return np.concatenate([axis * np.sin(half), np.cos(half)], axis=-1)


def skeleton(num_tracks, rng, style):
    """parent index per transform, parents first; style: chain | humanoid | star | forest"""
    parents = np.full(num_tracks, NO_PARENT, dtype=np.int32)
    for bone in range(1, num_tracks):
        if style == "chain":
            parents[bone] = bone - 1
        elif style == "star":
            parents[bone] = 0
        elif style == "forest":
            parents[bone] = NO_PARENT if bone % 7 == 0 else int(rng.integers(max(0, bone - 7), bone))
        else:                                   # humanoid: short chains hanging off recent bones
            parents[bone] = int(rng.integers(max(0, bone - 6), bone)) if rng.uniform() < 0.5 else bone - 1
    return parents


def raw_clip(spec):
    """[num_samples, num_tracks, 12] float32 (rotation xyzw | translation xyz_ | scale xyz_), parents, bind pose [num_tracks, 12]"""
    rng = np.random.default_rng(spec["seed"])
    num_tracks, num_samples, rate = spec["bones"], spec["samples"], spec["rate"]
    motion = spec.get("motion", "walk")
    parents = skeleton(num_tracks, rng, spec.get("skeleton", "humanoid"))
    t = np.arange(num_samples, dtype=np.float64) / rate
    duration = max(t[-1], 1.0 / rate) if num_samples > 1 else 1.0
    looping = spec.get("looping", False)

    # bind pose: a rest rotation and an offset from the parent per bone, unit scale
    bind = np.zeros((num_tracks, 12), dtype=np.float64)
    bind_rotation = quat_from_axis_angle(rng.normal(size=(num_tracks, 3)), rng.uniform(-0.8, 0.8, size=num_tracks))
    bind[:, 0:4] = bind_rotation
    bind[:, 4:7] = rng.normal(size=(num_tracks, 3)) * spec.get("bone_length", 12.0)
    bind[:, 8:11] = 1.0
    if spec.get("identity_bind", False):
        bind[:, 0:4] = [0, 0, 0, 1]
        bind[:, 4:7] = 0.0

    # a clip sub-track is default (== bind), constant (some other value) or animated
    def classes(default_p, constant_p):
        u = rng.uniform(size=num_tracks)
        return np.where(u < default_p, 0, np.where(u < default_p + constant_p, 1, 2))

    rotation_class = classes(spec.get("rotation_default", 0.05), spec.get("rotation_constant", 0.25))
    translation_class = classes(spec.get("translation_default", 0.3), spec.get("translation_constant", 0.45))
    scale_class = classes(spec.get("scale_default", 0.8), spec.get("scale_constant", 0.1)) if spec.get("scale", False) else np.zeros(num_tracks, dtype=np.int64)

    def periodic(shape, harmonics=3, amplitude=1.0):
        """smooth curves over the clip, [num_samples, *shape]; whole periods over the duration when the clip loops"""
        out = np.zeros((num_samples,) + shape)
        for h in range(1, harmonics + 1):
            cycles = h if looping else rng.uniform(0.3, 1.0) * h
            phase = rng.uniform(0, 2 * np.pi, size=shape)
            weight = rng.normal(size=shape) / h
            angle = 2 * np.pi * cycles * (t / (duration + (1.0 / rate if looping else 0.0)))
            out += weight * np.sin(angle.reshape((-1,) + (1,) * len(shape)) + phase)
        return out * amplitude

    rotation_amplitude = {"walk": 0.6, "hinge": 0.9, "half_turn": 0.25, "spin": 0.0, "tiny": 1e-4, "huge": 0.6, "still": 0.0}[motion]
    axis = rng.normal(size=(num_tracks, 3))
    if motion == "hinge":
        # rotation about ONE axis, the others idle with float noise: the grids of such clips hold values within 2^-47 of zero
        axis = np.eye(3)[rng.integers(0, 3, size=num_tracks)]
    angle = periodic((num_tracks,), amplitude=rotation_amplitude)
    if motion == "spin":
        angle = (rng.uniform(2.0, 9.0, size=num_tracks) * np.sign(rng.normal(size=num_tracks)))[None, :] * t[:, None]
        if looping and num_samples > 1:
            angle = 2 * np.pi * np.rint(rng.uniform(1, 3, size=num_tracks))[None, :] * (t[:, None] / (duration + 1.0 / rate))
    if motion == "half_turn":
        angle = angle + np.pi * (1.0 - 1e-3 * rng.uniform(size=num_tracks))[None, :]          # W = cos(angle / 2) near 0
    delta = quat_from_axis_angle(np.broadcast_to(axis, (num_samples, num_tracks, 3)).copy(), angle)
    if motion == "hinge":
        delta[..., 0:3] += rng.normal(size=delta[..., 0:3].shape) * 1e-7 * (np.abs(delta[..., 0:3]) < 1e-12)
        delta /= np.linalg.norm(delta, axis=-1, keepdims=True)
    constant_rotation = quat_from_axis_angle(rng.normal(size=(num_tracks, 3)), rng.uniform(-2.5, 2.5, size=num_tracks))
    rotation = quat_mul(np.broadcast_to(bind[:, 0:4], (num_samples, num_tracks, 4)), delta)
    rotation = np.where((rotation_class == 1)[None, :, None], quat_mul(bind[:, 0:4], constant_rotation)[None], rotation)
    rotation = np.where((rotation_class == 0)[None, :, None], bind[None, :, 0:4], rotation)

    translation_amplitude = {"walk": 4.0, "hinge": 0.5, "half_turn": 2.0, "spin": 1.0, "tiny": 1e-3, "huge": 1000.0, "still": 0.0}[motion]
    translation = bind[None, :, 4:7] + periodic((num_tracks, 3), amplitude=translation_amplitude)
    translation = np.where((translation_class == 1)[None, :, None], (bind[:, 4:7] + rng.normal(size=(num_tracks, 3)) * 3.0)[None], translation)
    translation = np.where((translation_class == 0)[None, :, None], bind[None, :, 4:7], translation)

    scale = np.ones((num_samples, num_tracks, 3))
    if spec.get("scale", False):
        animated_scale = 1.0 + periodic((num_tracks, 3), amplitude=0.25)
        if spec.get("uniform_scale", False):
            animated_scale = np.repeat(animated_scale[..., :1], 3, axis=-1)
        constant_scale = rng.uniform(0.5, 2.0, size=(num_tracks, 3))
        if spec.get("mirrored", False):
            constant_scale[:, 0] *= np.where(rng.uniform(size=num_tracks) < 0.5, -1.0, 1.0)      # a mirrored limb: negative X scale
        scale = np.where((scale_class == 2)[None, :, None], animated_scale, scale)
        scale = np.where((scale_class == 1)[None, :, None], constant_scale[None], scale)

    raw = np.zeros((num_samples, num_tracks, 12), dtype=np.float32)
    raw[..., 0:4] = rotation
    raw[..., 4:7] = translation
    raw[..., 8:11] = scale
    if looping and num_samples > 1:
        raw[-1] = raw[0]            # first == last: optimize_loops drops the last sample and sets the wrap policy
    return raw, parents, bind.astype(np.float32)


# ---- what to compress -------------------------------------------------------------------------------------------------------

VARIABLE = dict(rotation_format="quatf_drop_w_variable", translation_format="vector3f_variable", scale_format="vector3f_variable")
# /root/reference/test_data/configs/*.config.sjson (the four database configurations are DATABASES below)
CONFIGS = {
    "quant_medium": dict(level="medium", **VARIABLE),
    "quant_high": dict(level="high", **VARIABLE),
    "quant_highest": dict(level="highest", **VARIABLE),
    "raw": dict(level="medium", rotation_format="quatf_full", translation_format="vector3f_full", scale_format="vector3f_full"),
    "mixed_var_0": dict(level="medium", rotation_format="quatf_full", translation_format="vector3f_variable", scale_format="vector3f_variable"),
    "mixed_var_1": dict(level="medium", rotation_format="quatf_drop_w_variable", translation_format="vector3f_full", scale_format="vector3f_variable"),
    "drop_w_full": dict(level="medium", rotation_format="quatf_drop_w_full", translation_format="vector3f_variable", scale_format="vector3f_full"),      # (the third rotation format)
    "quant_mtx_error": dict(level="medium", matrix_error_metric=True, **VARIABLE),
    "quant_bind_relative": dict(level="medium", bind_relative=True, **VARIABLE),
    "keyframe_stripping": dict(level="medium", strip_proportion=0.0, strip_threshold=0.5, **VARIABLE),
}


def transform_specs():
    specs = []

    def add(name, config, **shape):
        spec = dict(name=f"{len(specs):03d}_{name}", config=config, seed=9000 + len(specs), rate=30.0)
        spec.update(shape)
        specs.append(spec)

    # every regression configuration over a spread of shapes (bones, samples, scale)
    shapes = [dict(bones=70, samples=91), dict(bones=33, samples=33, scale=True), dict(bones=5, samples=301), dict(bones=100, samples=64, scale=True, mirrored=True),
              dict(bones=16, samples=17, looping=True), dict(bones=160, samples=31), dict(bones=2, samples=600, scale=True), dict(bones=44, samples=120, motion="hinge")]
    for config in CONFIGS:
        for index, shape in enumerate(shapes):
            if config == "quant_highest" and shape["bones"] * shape["samples"] > 6000:
                shape = dict(shape, samples=min(shape["samples"], 48))           # (highest tries every permutation: kept small)
            add(f"{config}_{shape['bones']}x{shape['samples']}", config, **shape)
    # sample counts around the segmenting limits (<= 31 samples: one segment; ideal 16 / max 31 per segment), one and two samples
    for samples in (1, 2, 3, 4, 15, 16, 17, 30, 31, 32, 33, 34, 47, 48, 49, 63, 64, 65, 100, 301, 600):
        add(f"samples_{samples}", "quant_medium", bones=24, samples=samples, scale=samples % 2 == 0)
    for samples in (1, 2, 16, 31, 32, 33, 100):
        add(f"samples_{samples}_raw", "raw", bones=12, samples=samples, scale=samples % 2 == 1)
        add(f"samples_{samples}_mixed", "mixed_var_0" if samples % 2 else "mixed_var_1", bones=12, samples=samples, scale=True)
    # bone counts: groups of four rotations (SOA), sixteen sub-track types per word, pose windows of 104 tracks, Trooper_Main's 551
    for bones in (1, 2, 3, 4, 5, 7, 8, 9, 15, 16, 17, 31, 32, 33, 63, 64, 65, 103, 104, 105, 208, 209, 300, 551):
        add(f"bones_{bones}", "quant_medium", bones=bones, samples=20 if bones > 200 else 40, scale=bones % 3 == 0, skeleton=("chain", "humanoid", "star", "forest")[bones % 4])
    for bones in (1, 3, 4, 5, 16, 17, 105, 300):
        add(f"bones_{bones}_raw", "raw", bones=bones, samples=12, scale=bones % 2 == 0)
        add(f"bones_{bones}_drop_w_full", "drop_w_full", bones=bones, samples=35, scale=True)
    # motion that stresses the arithmetic
    for motion in ("hinge", "half_turn", "spin", "tiny", "huge", "still"):
        for config in ("quant_medium", "quant_high", "mixed_var_0"):
            add(f"motion_{motion}_{config}", config, bones=30, samples=50, motion=motion, scale=motion in ("huge", "spin"))
    # every sub-track constant / default / animated
    add("all_constant", "quant_medium", bones=20, samples=40, scale=True, rotation_default=0.0, rotation_constant=1.0, translation_default=0.0, translation_constant=1.0, scale_default=0.0, scale_constant=1.0)
    add("all_default", "quant_medium", bones=20, samples=40, rotation_default=1.0, translation_default=1.0, identity_bind=True)
    add("all_animated", "quant_medium", bones=20, samples=40, scale=True, rotation_default=0.0, rotation_constant=0.0, translation_default=0.0, translation_constant=0.0, scale_default=0.0, scale_constant=0.0)
    add("all_animated_raw", "raw", bones=20, samples=40, scale=True, rotation_default=0.0, rotation_constant=0.0, translation_default=0.0, translation_constant=0.0, scale_default=0.0, scale_constant=0.0)
    add("all_constant_raw", "raw", bones=20, samples=40, scale=True, rotation_default=0.0, rotation_constant=1.0, translation_default=0.0, translation_constant=1.0, scale_default=0.0, scale_constant=1.0)
    add("one_animated_rotation", "quant_medium", bones=40, samples=60, rotation_default=0.5, rotation_constant=0.475, translation_default=1.0)
    add("uniform_scale", "quant_medium", bones=25, samples=45, scale=True, uniform_scale=True, scale_default=0.2, scale_constant=0.2)
    # loop optimisation (wrap policy), with and without segments, with scale, stripped
    for samples in (2, 17, 33, 91):
        add(f"looping_{samples}", "quant_medium", bones=18, samples=samples, looping=True, scale=samples == 33)
        add(f"looping_{samples}_raw", "raw", bones=9, samples=samples, looping=True)
    add("looping_stripped", "keyframe_stripping", bones=18, samples=80, looping=True)
    # keyframe stripping: proportion, threshold, trivial only
    for proportion, threshold, trivial in ((0.3, 0.0, False), (0.6, 0.0, False), (0.0, 0.05, False), (0.0, 0.0, True), (0.9, 0.0, True)):
        add(f"strip_p{proportion}_t{threshold}_{int(trivial)}", "quant_medium", bones=28, samples=150, strip_proportion=proportion, strip_threshold=threshold, strip_trivial=trivial, scale=trivial)
    # precision, shell distance, sample rate
    for precision, shell in ((0.01, 3.0), (0.001, 1.0), (0.000001, 1.0), (0.1, 100.0)):
        add(f"precision_{precision}_shell_{shell}", "quant_medium", bones=30, samples=70, precision=precision, shell_distance=shell)
    for rate in (19.5, 24.0, 60.0, 120.0, 1.0):
        add(f"rate_{rate}", "quant_medium", bones=14, samples=77, rate=rate)
    # optional metadata behind the compressed data (the decoder must not care; aclhip_set_clip_hierarchy_from_metadata reads it)
    add("metadata_parents", "quant_medium", bones=40, samples=50, scale=True, include_parent_track_indices=True)
    add("metadata_descriptions", "quant_medium", bones=40, samples=50, include_track_descriptions=True, bind_defaults=True)
    add("metadata_everything", "quant_medium", bones=23, samples=33, scale=True, include_track_descriptions=True, include_parent_track_indices=True, include_track_names=True,
        include_track_list_name=True, include_contributing_error=True, bind_defaults=True)
    add("metadata_raw", "raw", bones=23, samples=33, include_track_descriptions=True, include_track_names=True, bind_defaults=True)
    add("bind_defaults_no_metadata", "quant_medium", bones=30, samples=40, scale=True, bind_defaults=True)
    return specs


# the four database configurations: (max_chunk_size, medium, low) -- uniformly_sampled_database{,_4kb,_4kb_mixed,_mixed}.config.sjson
DATABASES = {
    "database": dict(max_chunk_size=1024 * 1024, medium_proportion=0.0, low_proportion=0.5),
    "database_4kb": dict(max_chunk_size=4096, medium_proportion=0.0, low_proportion=0.5),
    "database_4kb_mixed": dict(max_chunk_size=4096, medium_proportion=0.3, low_proportion=0.4),
    "database_mixed": dict(max_chunk_size=1024 * 1024, medium_proportion=0.3, low_proportion=0.4),
}
DATABASE_CLIPS = [dict(bones=30, samples=150), dict(bones=12, samples=33, scale=True), dict(bones=70, samples=64), dict(bones=5, samples=301, looping=True),
                  dict(bones=104, samples=40, scale=True), dict(bones=18, samples=17), dict(bones=40, samples=220, motion="hinge"), dict(bones=3, samples=600)]


def compress(spec, **extra):
    raw, parents, bind = raw_clip(spec)
    config = dict(CONFIGS[spec["config"]])
    bind_relative = config.pop("bind_relative", False)
    options = dict(config)
    for key in ("strip_proportion", "strip_threshold", "strip_trivial", "precision", "shell_distance", "include_parent_track_indices", "include_track_descriptions",
                "include_track_names", "include_track_list_name", "include_contributing_error"):
        if key in spec:
            options[key] = spec[key]
    options.setdefault("precision", 0.01)
    options.setdefault("shell_distance", 3.0)
    options["optimize_loops"] = bool(spec.get("looping", False))
    options.update(extra)
    # bind pose relative (acl_compressor -bind_rel): the clip holds deltas from the bind pose, whose default sub-tracks are the identity;
    # bind_defaults: track_desc_transformf::default_value = the bind pose, sub-tracks that sit on it become DEFAULT sub-tracks
    use_bind = bool(spec.get("bind_defaults", False))
    if bind_relative:
        inverse = bind[:, 0:4] * np.array([-1, -1, -1, 1], dtype=np.float32)
        raw = raw.copy()
        raw[..., 0:4] = quat_mul(np.broadcast_to(inverse, raw[..., 0:4].shape).astype(np.float64), raw[..., 0:4].astype(np.float64)).astype(np.float32)
        raw[..., 4:7] -= bind[None, :, 4:7]
    blob = ob.ref_compress_ex(raw, spec["rate"], parents=parents, bind_pose=bind if use_bind else None, **options)
    return blob, parents, bind, use_bind


def main():
    if not ob.have_ref_compressor() or not ob.have_ref_database():
        raise SystemExit("oracle/_ref is missing: run `make -C oracle ref` where /root/reference exists")
    os.makedirs(OUT, exist_ok=True)
    started = time.time()

    specs = transform_specs()
    blobs, parents_list, binds, uses_bind = [], [], [], []
    for spec in specs:
        t0 = time.time()
        blob, parents, bind, use_bind = compress(spec)
        assert ob.ref().aclref_is_valid(blob.ctypes.data, 1) == 0, spec["name"]
        blobs.append(np.asarray(blob)), parents_list.append(parents), binds.append(bind), uses_bind.append(use_bind)
        print(f"{spec['name']}: {blob.size} bytes, {time.time() - t0:.1f} s", flush=True)
    offsets = np.cumsum([0] + [b.size for b in blobs]).astype(np.int64)
    track_offsets = np.cumsum([0] + [p.size for p in parents_list]).astype(np.int64)
    path = os.path.join(OUT, "transforms.npz")
    np.savez_compressed(path, blobs=np.concatenate(blobs), offsets=offsets, track_offsets=track_offsets, parents=np.concatenate(parents_list),
                        bind_poses=np.concatenate(binds), bind_is_default=np.array(uses_bind, dtype=np.uint8), specs=np.array(json.dumps(specs)))
    print(f"transforms.npz: {len(specs)} clips, {offsets[-1]} bytes of blobs, {os.path.getsize(path)} on disk")

    for name, build_options in DATABASES.items():
        clip_blobs = []
        for index, shape in enumerate(DATABASE_CLIPS):
            spec = dict(name=f"{name}_{index}", config="quant_medium", seed=12000 + 100 * len(clip_blobs) + index, rate=30.0, **shape)
            clip_blobs.append(compress(spec, enable_database_support=True)[0])
        ref = ob.ReferenceDatabase(clip_blobs, **build_options)
        clip_offsets = np.cumsum([0] + [clip.size for clip in ref.clips]).astype(np.int64)
        path = os.path.join(OUT, f"{name}.npz")
        np.savez_compressed(path, clips=np.concatenate([np.asarray(c) for c in ref.clips]), clip_offsets=clip_offsets, database=np.asarray(ref.database),
                            bulk_medium=np.asarray(ref.bulk[1]), bulk_low=np.asarray(ref.bulk[2]), options=np.array(json.dumps(build_options)))
        print(f"{name}.npz: {len(clip_blobs)} clips, chunks {ref.num_chunks}, {os.path.getsize(path)} on disk")
        ref.close()
    print(f"done in {time.time() - started:.0f} s")


if __name__ == "__main__":
    main()
