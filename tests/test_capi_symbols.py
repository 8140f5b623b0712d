"""The C-ABI shared library loads and exports every entry point include/aclhip.h declares (no compute calls, no GPU)."""
import ctypes
import os
import re

from acl_amd import runtime, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    header = open(os.path.join(ROOT, "include", "aclhip.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    return sorted(set(re.findall(r"\b(aclhip_[a-z_0-9]+)\s*\(", header)))


def test_header_declares_the_expected_surface():
    names = declared_functions()
    for required in ("aclhip_create", "aclhip_destroy", "aclhip_register_clip", "aclhip_unregister_clip",
                     "aclhip_decompress_tracks_batch", "aclhip_decompress_track_batch"):
        assert required in names


def test_library_exports_every_declared_symbol():
    lib = runtime.load_library()
    missing = [name for name in declared_functions() if not hasattr(lib, name)]
    assert not missing, missing
    assert sorted(runtime.EXPORTED_SYMBOLS) == declared_functions()


def test_default_params_match_the_reference_defaults():
    params = runtime.default_params()
    # default_transform_decompression_settings (decompression_settings.h:211-232) + track_writer defaults (track_writer.h:161-163)
    assert params.rounding_policy == runtime.ROUND_NONE
    assert params.looping_policy == runtime.LOOP_AS_COMPRESSED
    assert params.normalization == runtime.NORMALIZE_LERP_ONLY
    assert params.per_track_rounding == 0
    assert (params.default_rotation_mode, params.default_translation_mode, params.default_scale_mode) == (runtime.DEFAULT_CONSTANT, runtime.DEFAULT_CONSTANT, runtime.DEFAULT_LEGACY)


def test_status_strings_and_argument_checks_without_a_device():
    lib = runtime.load_library()
    assert lib.aclhip_status_string(0) == b"ok"
    assert lib.aclhip_status_string(2) == b"invalid compressed_tracks"
    # null out pointer is rejected before any HIP call
    assert lib.aclhip_create(0, None) == 1


def test_synth_library_loads():
    spec = synth.default_spec()
    assert spec.num_tracks == 100 and spec.num_samples == 301


def test_product_never_imports_the_oracle():
    """The oracle is test infrastructure: nothing under acl_amd/ or include/ may reference it."""
    offenders = []
    for base in ("acl_amd", "include"):
        for dirpath, _, filenames in os.walk(os.path.join(ROOT, base)):
            for filename in filenames:
                if filename == "build.py":
                    continue    # building the checker (make -C oracle) is not using it
                if filename.endswith((".py", ".h", ".hpp", ".cpp", ".hip")):
                    text = open(os.path.join(dirpath, filename), errors="ignore").read()
                    if re.search(r"from oracle|import oracle|acl_oracle\.h|libacloracle|libaclref|oracle/", text):
                        offenders.append(os.path.join(dirpath, filename))
    assert not offenders, offenders


def test_header_is_plain_c_and_host_entry_points_work_without_a_gpu(tmp_path):
    """include/aclhip.h from a C99 translation unit (gcc -std=c99 -pedantic -Werror), linked against libaclhip.so, run on this CPU box."""
    import subprocess
    import numpy as np
    from acl_amd import synth
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib_dir = os.path.dirname(runtime.library_path())
    binary = tmp_path / "abi_smoke"
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", os.path.join(root, "include"), os.path.join(root, "tests", "c", "abi_smoke.c"),
                    "-L" + lib_dir, "-laclhip", "-Wl,-rpath," + lib_dir, "-o", str(binary)], check=True)
    clip = synth.build_clip(seed=8, num_tracks=12, num_samples=40)
    valid, broken = tmp_path / "valid.acl", tmp_path / "broken.acl"
    clip.blob.tofile(valid)
    corrupt = clip.blob.copy()
    corrupt[50] ^= 0x5A
    corrupt.tofile(broken)
    assert subprocess.run([str(binary), str(valid), "valid"]).returncode == 0
    assert subprocess.run([str(binary), str(broken), "invalid"]).returncode == 0
    # the call sequences of INTEGRATION.md: must compile as C99 and link with the signatures the document shows
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-shared", "-fPIC", "-I", os.path.join(root, "include"),
                    os.path.join(root, "tests", "c", "integration_example.c"), "-L" + lib_dir, "-laclhip", "-Wl,--no-undefined", "-o", str(tmp_path / "libintegration_example.so")], check=True)


def test_cpp_mirror_and_its_drivers_compile_warning_free():
    """acl_amd/csrc/aclhip.hpp (the C++ mirror of decompression_context / database_context / track_writer) and the
    programs that drive it on the GPU box build with g++ -Wall -Wextra -Werror here (they run under -m gpu)."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for source in ("context_mirror_test.cpp", "database_mirror_test.cpp", "scalar_mirror_test.cpp", "pose_consumers_mirror_test.cpp"):
        subprocess.run(["g++", "-std=c++17", "-Wall", "-Wextra", "-Werror", "-fsyntax-only", os.path.join(root, "tests", "cpp", source)], check=True)


def test_header_library_and_binding_agree_on_the_abi_version():
    """a caller built against another include/aclhip.h hands over structs of another shape (found in round 3: a stale adapter binary
    passed an aclhip_output_desc without `skip_tracks`): the header carries a version, the library reports the one it was built with"""
    header = open(os.path.join(ROOT, "include", "aclhip.h")).read()
    declared = int(re.search(r"#define\s+ACLHIP_ABI_VERSION\s+(\d+)u", header).group(1))
    lib = runtime.load_library()
    assert lib.aclhip_abi_version() == declared == runtime.ABI_VERSION
    # the binding's mirrors of the structs that changed last (ABI 5): layout u32 | 3 skip bytes + 1 reserved | rows | skip_tracks |
    # mask_table | instance_masks | instance_track_counts | mask_stride u32 + 1 reserved; ... | instance_looping_policies
    assert ctypes.sizeof(runtime.OutputDesc) == 56
    assert runtime.OutputDesc.skip_tracks.offset == 16 and runtime.OutputDesc.mask_table.offset == 24 and runtime.OutputDesc.mask_stride.offset == 48
    assert ctypes.sizeof(runtime.DecompressParams) == 64 and runtime.DecompressParams.instance_looping_policies.offset == 32
    assert runtime.DecompressParams.track_rounding_table.offset == 40 and runtime.DecompressParams.track_rounding_stride.offset == 56
