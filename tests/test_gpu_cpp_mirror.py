"""The C++ mirror of acl::decompression_context (acl_amd/csrc/aclhip.hpp) built with g++ against the C-ABI library and driven
like the reference's validator. Needs a GPU."""
import os
import subprocess

import numpy as np
import pytest

from acl_amd import runtime, synth
from oracle import bindings as ob

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def mirror_binary(tmp_path_factory):
    out = tmp_path_factory.mktemp("cpp") / "context_mirror_test"
    lib_dir = os.path.dirname(runtime.library_path())
    subprocess.run(["g++", "-std=c++17", "-O1", "-Wall", "-Wextra", os.path.join(ROOT, "tests", "cpp", "context_mirror_test.cpp"),
                    "-L" + lib_dir, "-laclhip", "-Wl,-rpath," + lib_dir, "-o", str(out)], check=True)
    return str(out)


@pytest.mark.parametrize("mode", ["identity", "skipped", "variable"])
def test_cpp_context_matches_oracle(mirror_binary, tmp_path, mode):
    clip = synth.build_clip(seed=31, num_tracks=23, num_samples=80, has_scale=1, rotation_default=0.2, translation_default=0.3, scale_default=0.5)
    blob_path, times_path, out_path = tmp_path / "clip.acl", tmp_path / "times.txt", tmp_path / "poses.bin"
    clip.blob.tofile(blob_path)
    rng = np.random.default_rng(5)
    times = rng.uniform(0.0, clip.duration, size=20).astype(np.float32)
    times_path.write_text("\n".join(repr(float(t)) for t in times))

    result = subprocess.run([mirror_binary, str(blob_path), str(times_path), str(out_path), mode])
    assert result.returncode == 0

    poses = np.fromfile(out_path, dtype=np.float32).reshape(times.size, clip.num_tracks, 12)
    defaults = None
    if mode == "variable":
        defaults = np.zeros((clip.num_tracks, 12), dtype=np.float32)
        for i in range(clip.num_tracks):
            defaults[i] = [0.5, -0.5, 0.5, 0.5 + i, i, 2.0, 3.0, 0.0, 2.0, i, 2.0, 0.0]
    options = {
        "identity": ob.default_options(),
        "skipped": ob.default_options(default_rotation_mode=ob.DEFAULT_SKIPPED, default_translation_mode=ob.DEFAULT_SKIPPED, default_scale_mode=ob.DEFAULT_SKIPPED),
        "variable": ob.default_options(default_rotation_mode=ob.DEFAULT_VARIABLE, default_translation_mode=ob.DEFAULT_VARIABLE, default_scale_mode=ob.DEFAULT_VARIABLE),
    }[mode]
    if defaults is not None:
        options.default_values = defaults.ctypes.data
    lanes = [0, 1, 2, 3, 4, 5, 6, 8, 9, 10]
    for i, t in enumerate(times):
        expected = np.full((clip.num_tracks, 12), -7.0, dtype=np.float32)
        ob.oracle_decompress_tracks(clip.blob, float(t), 0, options, out=expected)
        assert np.array_equal(poses[i][:, lanes].view(np.uint32), expected[:, lanes].view(np.uint32))


def test_cpp_context_full_formats_and_metadata(mirror_binary, tmp_path):
    """a quatf_full + vector3f_full clip with track descriptions from the reference's compressor (tests/golden/corpus): the default settings
    refuse it, settings that support every format decode it to the oracle's bits, its metadata reads like the reference's accessors
    return it, and decompress_pose(object space) takes the skeleton from the blob"""
    import helpers
    clip = next(clip for clip in helpers.load_corpus() if clip["name"].endswith("metadata_raw"))
    blob_path, times_path, out_path = tmp_path / "clip.acl", tmp_path / "times.txt", tmp_path / "poses.bin"
    clip["blob"].tofile(blob_path)
    times, _ = helpers.corpus_sample_times(clip["blob"])
    times = times[::3]
    times_path.write_text("\n".join(repr(float(t)) for t in times))
    result = subprocess.run([mirror_binary, str(blob_path), str(times_path), str(out_path), "formats"])
    assert result.returncode == 0
    bones = clip["spec"]["bones"]
    raw = np.fromfile(out_path, dtype=np.float32)
    parents = raw[:bones].view(np.uint32)
    assert np.array_equal(parents, clip["parents"].astype(np.uint32))
    defaults = raw[bones: bones + bones * 12].reshape(bones, 12)
    expected_defaults = clip["bind_pose"].copy()
    expected_defaults[:, 7] = 0.0
    expected_defaults[:, 11] = 0.0
    assert np.array_equal(defaults.view(np.uint32), expected_defaults.view(np.uint32))
    poses = raw[bones + bones * 12:].reshape(times.size, 2, bones, 12)
    options = helpers.oracle_options(4)         # default settings that take every packed format (oracle/ref_bridge.cpp: any_format_settings)
    lanes = [0, 1, 2, 3, 4, 5, 6, 8, 9, 10]
    for i, t in enumerate(times):
        local = ob.oracle_decompress_tracks(clip["blob"], float(t), 0, options)
        assert np.array_equal(poses[i, 0][:, lanes].view(np.uint32), local[:, lanes].view(np.uint32))
        in_object_space = ob.oracle_local_to_object_space(clip["parents"].astype(np.uint32), local)
        assert np.array_equal(poses[i, 1][:, lanes].view(np.uint32), in_object_space[:, lanes].view(np.uint32))


@pytest.fixture(scope="module")
def database_mirror_binary(tmp_path_factory):
    out = tmp_path_factory.mktemp("cpp") / "database_mirror_test"
    lib_dir = os.path.dirname(runtime.library_path())
    subprocess.run(["g++", "-std=c++17", "-O1", "-Wall", "-Wextra", os.path.join(ROOT, "tests", "cpp", "database_mirror_test.cpp"),
                    "-L" + lib_dir, "-laclhip", "-Wl,-rpath," + lib_dir, "-o", str(out)], check=True)
    return str(out)


@pytest.mark.parametrize("name", ["three_clips_4k_chunks", "medium_tier_only"])
def test_cpp_database_context_matches_oracle(database_mirror_binary, tmp_path, name):
    import helpers
    from oracle.database import OracleDatabase
    case = helpers.load_database_golden(name)
    paths = {key: tmp_path / f"{key}.bin" for key in ("database", "bulk_medium", "bulk_low", "clip")}
    for key in ("database", "bulk_medium", "bulk_low"):
        case[key].tofile(paths[key])
    clip_index = len(case["clips"]) - 1
    clip = case["clips"][clip_index]
    clip.tofile(paths["clip"])
    times = case["times"][clip_index]
    (tmp_path / "times.txt").write_text("\n".join(repr(float(t)) for t in times))
    out_path = tmp_path / "poses.bin"

    result = subprocess.run([database_mirror_binary, str(paths["database"]), str(paths["bulk_medium"]), str(paths["bulk_low"]), str(paths["clip"]),
                             str(tmp_path / "times.txt"), str(out_path)])
    assert result.returncode == 0

    num_tracks = ob.oracle().aclo_num_tracks(clip.ctypes.data)
    poses = np.fromfile(out_path, dtype=np.float32).reshape(4, times.size, num_tracks, 12)
    oracle_db = OracleDatabase(case["database"], case["bulk_medium"], case["bulk_low"])
    lanes = [0, 1, 2, 3, 4, 5, 6]
    for state in range(4):
        if state == 1:
            oracle_db.stream_in(1)
        elif state == 2:
            oracle_db.stream_in(2)
        elif state == 3:
            oracle_db.stream_out(1)
            oracle_db.stream_out(2)
        for i, t in enumerate(times):
            expected = oracle_db.decompress_tracks(clip, float(t))
            assert np.array_equal(poses[state, i][:, lanes].view(np.uint32), expected[:, lanes].view(np.uint32)), (state, i)
    if name == "three_clips_4k_chunks":
        assert not np.array_equal(poses[0], poses[2])     # the tiers made a difference
    assert np.array_equal(poses[0], poses[3])


@pytest.fixture(scope="module")
def scalar_mirror_binary(tmp_path_factory):
    out = tmp_path_factory.mktemp("cpp") / "scalar_mirror_test"
    lib_dir = os.path.dirname(runtime.library_path())
    subprocess.run(["g++", "-std=c++17", "-O1", "-Wall", "-Wextra", os.path.join(ROOT, "tests", "cpp", "scalar_mirror_test.cpp"),
                    "-L" + lib_dir, "-laclhip", "-Wl,-rpath," + lib_dir, "-o", str(out)], check=True)
    return str(out)


@pytest.mark.parametrize("track_type", [0, 1, 2, 3, 4])
def test_cpp_context_on_scalar_track_lists(scalar_mirror_binary, tmp_path, track_type):
    clip = synth.build_scalar_clip(seed=40 + track_type, track_type=track_type, num_tracks=14, num_samples=30)
    blob_path, times_path, out_path = tmp_path / "clip.acl", tmp_path / "times.txt", tmp_path / "values.bin"
    clip.blob.tofile(blob_path)
    rng = np.random.default_rng(6)
    times = rng.uniform(0.0, clip.duration, size=12).astype(np.float32)
    times_path.write_text("\n".join(repr(float(t)) for t in times))
    result = subprocess.run([scalar_mirror_binary, str(blob_path), str(times_path), str(out_path)])
    assert result.returncode == 0
    values = np.fromfile(out_path, dtype=np.float32).reshape(times.size, 2, clip.num_tracks, clip.num_components)
    for i, t in enumerate(times):
        expected = ob.oracle_scalar_decompress_tracks(clip.blob, float(t))
        assert np.array_equal(values[i, 0].view(np.uint32), expected.view(np.uint32))
        assert np.array_equal(values[i, 1].view(np.uint32), expected.view(np.uint32))     # decompress_track == decompress_tracks


@pytest.fixture(scope="module")
def pose_consumers_mirror_binary(tmp_path_factory):
    out = tmp_path_factory.mktemp("cpp") / "pose_consumers_mirror_test"
    lib_dir = os.path.dirname(runtime.library_path())
    subprocess.run(["g++", "-std=c++17", "-O1", "-Wall", "-Wextra", os.path.join(ROOT, "tests", "cpp", "pose_consumers_mirror_test.cpp"),
                    "-L" + lib_dir, "-laclhip", "-Wl,-rpath," + lib_dir, "-o", str(out)], check=True)
    return str(out)


@pytest.mark.parametrize("name", ["biped_40_scale", "rig_100_two_roots"])
def test_cpp_context_pose_consumers(pose_consumers_mirror_binary, tmp_path, name):
    """decompress_pose of the C++ mirror == the oracle's decode -> apply_additive_to_base -> local_to_object_space, bit for bit"""
    import helpers
    case = helpers.load_consumer_golden(name)
    paths = {key: tmp_path / key for key in ("additive.acl", "base.acl", "parents.bin", "times.txt", "poses.bin")}
    case["additive_blob"].tofile(paths["additive.acl"])
    case["base_blob"].tofile(paths["base.acl"])
    case["parents"].astype(np.uint32).tofile(paths["parents.bin"])
    times = case["times"][:5]
    paths["times.txt"].write_text("\n".join(f"{float(a)!r} {float(b)!r}" for a, b in times))
    result = subprocess.run([pose_consumers_mirror_binary] + [str(paths[key]) for key in ("additive.acl", "base.acl", "parents.bin", "times.txt", "poses.bin")])
    assert result.returncode == 0
    num_tracks = case["parents"].size
    everything = np.fromfile(paths["poses.bin"], dtype=np.float32)
    poses = everything[: times.shape[0] * 8 * num_tracks * 12].reshape(times.shape[0], 4, 2, num_tracks, 12)
    blended = everything[times.shape[0] * 8 * num_tracks * 12:].reshape(2, num_tracks, 12)
    expected_blend = ob.oracle_blend_poses([ob.oracle_decompress_tracks(case["additive_blob"], float(times[0][0])), ob.oracle_decompress_tracks(case["base_blob"], float(times[0][1]))], [0.25, 0.75])
    assert helpers.exact(blended[0], expected_blend)
    assert helpers.exact(blended[1], ob.oracle_local_to_object_space(case["parents"], expected_blend))
    for i, (additive_time, base_time) in enumerate(times):
        additive_pose = ob.oracle_decompress_tracks(case["additive_blob"], float(additive_time))
        base_pose = ob.oracle_decompress_tracks(case["base_blob"], float(base_time))
        for additive_format in range(4):
            local = ob.oracle_apply_additive_to_base(additive_format, base_pose, additive_pose)
            assert helpers.exact(poses[i, additive_format, 0], local)
            assert helpers.exact(poses[i, additive_format, 1], ob.oracle_local_to_object_space(case["parents"], local))
