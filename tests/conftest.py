import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """Tests marked `gpu` are skipped (not failed) on a box without a device, so a plain `pytest tests` is green there too."""
    try:
        import torch
        have_gpu = torch.cuda.is_available()
    except ImportError:
        have_gpu = False
    if have_gpu:
        return
    skip = pytest.mark.skip(reason="needs a GPU (MI355X): run with -m gpu on the GPU box")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session", autouse=True)
def native_libraries():
    """Builds (or reuses) the in-tree native libraries: libaclhip.so, libaclsynth.so, the CPU oracle."""
    from acl_amd import build
    paths = build.build_all()
    # PyTorch bundles its own HIP runtime; when a process uses both torch and libaclhip.so (GPU tests allocate device memory with
    # torch), torch has to be imported FIRST or it finds "no HIP GPUs" -- bench.py and __graft_entry__.smoke() do the same.
    try:
        import torch
        if torch.cuda.is_available():
            torch.cuda.init()
    except ImportError:
        pass
    return paths


# Clip shapes exercised by both the CPU (oracle vs reference) and the GPU (kernel vs oracle) parity tests.
# Edge cases follow what the reference's regression validator covers (tools/acl_compressor/sources/validate_tracks.cpp):
# single / multi segment, scale, wrap looping, stripped keyframes, old format version, one or two samples, raw and constant rates.
CLIP_SPECS = {
    "cmu_100": dict(),
    "cmu_70_default": dict(seed=7, num_tracks=70),
    "single_segment": dict(num_samples=20),
    "scale_37": dict(has_scale=1, num_tracks=37),
    "stripped": dict(strip_keyframes=1),
    "wrap_77": dict(wrap=1, num_samples=77),
    "v2_0_low_bits": dict(version=7, min_bits=3, max_bits=19),
    "v2_1_wip": dict(version=8, min_bits=3, max_bits=19, raw_fraction=0.1),
    "stripped_single_segment_scale": dict(strip_keyframes=1, num_samples=25, has_scale=1),
    "stripped_wrap_scale": dict(strip_keyframes=1, wrap=1, num_samples=100, has_scale=1, num_tracks=19),
    "one_sample": dict(num_samples=1),
    "two_samples_three_tracks": dict(num_samples=2, num_tracks=3),
    "raw_and_constant_rates": dict(raw_fraction=0.3, width0_fraction=0.3, num_tracks=64, has_scale=1, scale_default=0.2, scale_constant=0.2, translation_constant=0.3),
    "cinematic_300": dict(num_tracks=300, has_scale=1, scale_default=0.5, scale_constant=0.1, rotation_constant=0.2, translation_constant=0.3, num_samples=200),
    "three_full_windows_320": dict(num_tracks=320, num_samples=40, has_scale=1, scale_default=0.3, scale_constant=0.3, raw_fraction=0.05),     # 960 quads = 3 x 320
    "crowd_rig_1200": dict(num_tracks=1200, num_samples=33, rotation_constant=0.3, translation_constant=0.6, wrap=1),
    "giant_2500_all_animated": dict(num_tracks=2500, num_samples=6, rotation_default=0.0, rotation_constant=0.0, translation_default=0.0, translation_constant=0.0),
    "all_default": dict(num_tracks=17, rotation_default=1.0, translation_default=1.0),
    "all_animated_65": dict(num_tracks=65, rotation_default=0.0, rotation_constant=0.0, translation_default=0.0, translation_constant=0.0, num_samples=40),
    "max_segment_31": dict(num_samples=31, num_tracks=9),
    "two_segments_32": dict(num_samples=32, num_tracks=9),
    "high_bits_23": dict(min_bits=20, max_bits=23, num_tracks=33, num_samples=50),
    "default_scale_zero": dict(has_scale=1, default_scale=0, num_tracks=12, num_samples=10),
}


def sample_times_for(duration, count, rng):
    """Random times slightly outside [0, duration] (clamping) plus the exact ends and the middle."""
    times = rng.uniform(-0.1, duration + 0.1, size=count).astype(np.float32)
    return np.concatenate([times, np.array([0.0, duration, duration * 0.5, -1.0, duration + 1.0], dtype=np.float32)])


@pytest.fixture(scope="session")
def clip_specs():
    return CLIP_SPECS


def random_clip_specs(count, seed):
    """Seeded random clip shapes for the sweep tests: track counts around the 16-track type word and 4-rotation group boundaries,
    sample counts around the 16/17/32-sample segment boundaries, every class mix, both format generations."""
    rng = np.random.default_rng(seed)
    specs = []
    for index in range(count):
        has_scale = int(rng.uniform() < 0.4)
        rotation_default = float(rng.choice([0.0, 0.05, 0.3]))
        translation_default = float(rng.choice([0.0, 0.05, 0.5]))
        spec = dict(
            seed=int(1000 + seed * 100 + index),
            num_tracks=int(rng.choice([1, 2, 3, 4, 5, 15, 16, 17, 31, 32, 33, 47, 63, 64, 65, 100, 107, 130])),
            num_samples=int(rng.choice([1, 2, 3, 15, 16, 17, 18, 31, 32, 33, 34, 47, 48, 49, 64, 65, 100, 129])),
            sample_rate=float(rng.choice([24.0, 30.0, 60.0, 120.0])),
            version=int(rng.choice([7, 8, 9, 10])),
            has_scale=has_scale,
            default_scale=int(rng.integers(0, 2)),
            wrap=int(rng.uniform() < 0.3),
            strip_keyframes=int(rng.uniform() < 0.3),
            strip_fraction=float(rng.uniform(0.1, 0.6)),
            rotation_default=rotation_default,
            rotation_constant=float(rng.uniform(0.0, 1.0 - rotation_default)),
            translation_default=translation_default,
            translation_constant=float(rng.uniform(0.0, 1.0 - translation_default)),
            scale_default=0.3 if has_scale else 0.0,
            scale_constant=0.3 if has_scale else 0.0,
            min_bits=int(rng.integers(1, 9)),
            max_bits=int(rng.integers(9, 24)),
            width0_fraction=float(rng.choice([0.0, 0.05, 0.3])),
            raw_fraction=float(rng.choice([0.0, 0.02, 0.2])),
        )
        if spec["version"] == 7:
            spec["strip_keyframes"] = 0       # keyframe stripping appeared with v02_01_99
        specs.append(spec)
    return specs
