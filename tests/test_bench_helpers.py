"""bench.py's host side arithmetic (no GPU): the floors that decide an entry's `bound`, the workload builder's shapes (BASELINE.json
configs), the default run's spec list and its traffic keys, the CPU topology reader."""
import json
import os

import numpy as np

import bench

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bound_is_the_higher_of_the_two_floors():
    # the headline batch: 314.7 MB at 8 TB/s = 39.3 us; 17.7 M vector instructions x 4 cycles / 1 024 SIMDs / 2.4 GHz = 28.8 us
    floors = bench.bound_of(0.0512, 314_665_440, 17_688_730)
    assert floors["bound"] == "hbm"
    assert abs(floors["hbm_floor_ms"] - 0.039333) < 1e-5 and abs(floors["valu_issue_floor_ms"] - 0.028790) < 1e-5
    assert abs(floors["frac_of_bound"] - 0.039333 / 0.0512) < 1e-4
    # the 300-bone rig: 93.8 M instructions = 152.7 us of issue under a 198 us launch
    rig = bench.bound_of(0.198, 944_666_400, 93_840_182)
    assert rig["bound"] == "valu" and abs(rig["valu_issue_floor_ms"] - 0.15273) < 1e-4 and abs(rig["frac_of_bound"] - 0.15273 / 0.198) < 1e-3
    # without a counter pass the HBM floor stands alone
    alone = bench.bound_of(0.05, 314_665_440, None)
    assert alone["bound"] == "hbm" and alone["valu_issue_floor_ms"] is None and alone["valu_instructions"] is None


def test_workloads_have_the_shapes_baseline_json_names():
    baseline = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    assert "64k instances of one 100-bone clip" in baseline["configs"][1]
    clips, clip_indices, times = bench.build_workload("one_clip", 0, bench.INSTANCES_PER_GPU)
    assert len(clips) == 1 and clips[0].num_tracks == 100 and clip_indices.size == 65536 and clip_indices.max() == 0
    assert times.dtype == np.float32 and times.min() >= 0.0 and times.max() <= clips[0].duration
    clips, clip_indices, _ = bench.build_workload("256_clips", 0, 4096)
    assert len(clips) == 256 and all(c.num_tracks == 100 for c in clips) and len(set(clip_indices.tolist())) > 200
    clips, _, _ = bench.build_workload("cinematic", 0, 16)
    assert clips[0].num_tracks == 300
    # every rank draws its own shard
    _, _, times_a = bench.build_workload("one_clip", 0, 256)
    _, _, times_b = bench.build_workload("one_clip", 1, 256)
    assert not np.array_equal(times_a, times_b)


def test_default_run_specs_have_distinct_traffic_keys_and_known_workloads():
    specs = bench.default_run_specs()
    keys = [bench.spec_key(name, options) for name, options, _ in specs]
    assert len(set(keys)) == len(keys)
    assert keys[0] == "one_clip"                                      # the headline's own traffic
    assert all(name in bench.WORKLOAD_TEXT for name, _, _ in specs)
    for wanted in ("one_clip_lods", "256_clips, attached order", "track_requests", "one_clip, qv32", "database, list order"):
        assert wanted in keys, wanted
    assert bench.traffic_key_of("256_clips", "locality", "qvv48", keep_rows=True) is None


def test_cpu_topology_is_sane_or_absent():
    sockets, cores_per_socket, threads_per_core = bench.cpu_topology()
    if sockets is not None:
        assert sockets >= 1 and (cores_per_socket is None or cores_per_socket >= 1) and threads_per_core >= 1
        if cores_per_socket is not None:
            assert sockets * cores_per_socket * threads_per_core >= 1
