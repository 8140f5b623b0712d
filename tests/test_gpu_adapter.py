"""oracle/_ref/adapter_parity_test: acl::decompression_context (the reference's own headers) and acl_gpu::decompression_context
(acl_amd/csrc/acl_gpu_adapter.h over libaclhip.so) in ONE process behind the same interface, driven by the same acl::track_writer
types -- identity / skipped / variable defaults, every rounding policy incl. per track, the three looping policies, default and debug
settings. The binary is built where /root/reference exists (oracle/Makefile) and travels to the GPU box prebuilt."""
import os
import subprocess

import pytest

from acl_amd import synth
import helpers
from conftest import CLIP_SPECS

pytestmark = pytest.mark.gpu

BINARY = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref", "adapter_parity_test")


@pytest.mark.skipif(not os.path.exists(BINARY), reason="oracle/_ref/adapter_parity_test not built (needs /root/reference at build time)")
def test_reference_context_and_gpu_adapter_agree_bit_for_bit(tmp_path):
    paths = []
    for name in ("cmu_70_default", "scale_37", "stripped_wrap_scale", "raw_and_constant_rates", "two_segments_32", "one_sample", "v2_0_low_bits", "cinematic_300"):
        path = tmp_path / f"{name}.acl"
        synth.build_clip(**CLIP_SPECS[name]).blob.tofile(path)
        paths.append(str(path))
    for name in helpers.golden_cases():
        if name.startswith("real_"):            # written by the reference's own compressor
            path = tmp_path / f"{name}.acl"
            helpers.load_golden(name)["blob"].tofile(path)
            paths.append(str(path))
    for name in ("float1f_all_rates", "float3f_wrap", "vector4f_low_bits", "float2f_v2_0_rates"):
        path = tmp_path / f"{name}.acl"
        synth.build_scalar_clip(**helpers.SCALAR_CLIP_SPECS[name]).blob.tofile(path)
        paths.append(str(path))
    for name in helpers.scalar_golden_cases():      # scalar lists written by the reference's own compressor
        path = tmp_path / f"scalar_{name}.acl"
        helpers.load_scalar_golden(name)["blob"].tofile(path)
        paths.append(str(path))
    result = subprocess.run([BINARY] + paths, capture_output=True, text=True, timeout=600)
    assert result.returncode == 0, f"exit code {result.returncode}\n{result.stdout}\n{result.stderr}"
    assert result.stdout.count("bit identical") == len(paths)
