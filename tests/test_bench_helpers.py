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


def _round5_record():
    """the 26 KB line of round 5's default run (the one the driver could not parse), as committed"""
    return json.loads(open(os.path.join(ROOT, "profiles", "r05_bench.json")).read().strip().splitlines()[-1])


def test_headline_of_the_default_run_stays_under_4_kb():
    full = _round5_record()
    assert len(json.dumps(full)) > 20000
    line = bench.compact_headline(full)
    text = json.dumps(line)
    assert len(text) < 4096, len(text)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data"):
        assert key in line, key
    assert line["config"]["workload"].startswith("64k instances of one CMU-shaped 100-bone clip") and line["dtype"] == "f32"
    roofline = line["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic", "algorithmic_bytes_per_launch", "kernel", "kernel_ms", "valu_issue_floor_ms", "frac_of_bound"):
        assert key in roofline, key
    assert abs(roofline["frac"] - full["roofline"]["frac"]) < 1e-5 and roofline["traffic"] == full["roofline"]["traffic"]
    cpu = line["cpu_baseline"]
    for key in ("value", "unit", "cores", "threads_at_best", "nproc", "physical_cores", "per_thread_1t", "kind", "sample", "gpu_over_cpu", "gpu_over_cpu_extrapolated_physical_cores"):
        assert key in cpu, key
    assert line["self_check"] == {"instances": 256, "max_abs_err": 0.0, "bit_exact": True}
    # one row per extra workload: [kernel_ms, frac, bound, frac_of_bound, traffic / algorithmic bytes]
    assert line["workload_columns"] == ["kernel_ms", "frac", "bound", "frac_of_bound", "traffic_ratio"]
    assert "workloads_dropped_for_size" not in line
    rows = line["workloads"]
    for name in ("256_clips", "cinematic", "database", "track_requests", "scalar", "blend_object_space", "one_clip, qv32", "one_clip, 131072 instances", "database, paged on a second stream"):
        assert name in rows, name
    by_name = {entry["workload"]: entry for entry in full["workloads"]}
    assert abs(rows["cinematic"][0] - by_name["cinematic"]["kernel_ms"]) < 1e-4 and rows["cinematic"][2] == by_name["cinematic"]["bound"]
    assert abs(rows["256_clips"][4] - by_name["256_clips"]["traffic"] / by_name["256_clips"]["algorithmic_bytes"]) < 1e-3


def test_headline_of_an_8_gpu_run_stays_under_4_kb():
    full = _round5_record()
    full["n_gpus"] = 8
    full.pop("cpu_baseline"), full.pop("layouts"), full.pop("footprint_sweep")
    ranks = [0.0501 + 0.0001 * k for k in range(8)]
    full["roofline"]["kernel_ms_per_rank"] = ranks
    full["checks"] = {"backend": "nccl", "n_gpus_claimed": 8, "ok": False, "communicator_ranks": 8, "distinct_devices": 8, "kernel_ms_per_rank": ranks, "kernel_ms_min": min(ranks),
                      "kernel_ms_max": max(ranks), "devices": [f"host/0000:{k:02x}:00" for k in range(8)], "peer_access": ["7/7"] * 8, "problems": ["x" * 400] * 9}
    gather = {"status": "done", "shard_bytes": 314572800, "rccl_all_gather_ms": 21.5, "rccl_all_gather_gbps_into_each_rank": 102.4, "p2p_to_rank0_ms": 2.9, "p2p_gbps_into_rank0": 760.0,
              "p2p_gbps_per_link": 108.5, "p2p_link_frac_of_153_gbps": 0.709, "p2p_error": "e" * 300}
    full["gather"] = gather
    full["workloads"] = [{"workload": name, "config": "c" * 200, "n_gpus": 8, "kernel_ms": 0.19, "frac": 0.6, "algorithmic_bytes_per_gpu": 944666400, "gather": gather}
                         for name in ("cinematic", "database")] + [{"workload": "broken", "error": "z" * 300}]
    line = bench.compact_headline(full)
    assert len(json.dumps(line)) < 4096
    assert line["n_gpus"] == 8 and len(line["roofline"]["kernel_ms_per_rank"]) == 8 and line["checks"]["communicator_ranks"] == 8
    assert len(line["checks"]["problems"]) == 4 and all(len(problem) <= 120 for problem in line["checks"]["problems"])
    assert line["gather"]["p2p_to_rank0_ms"] == 2.9 and len(line["gather"]["p2p_error"]) == 120
    assert set(line["workloads"]) == {"cinematic", "database", "broken"} and line["workloads"]["broken"][2] == "error"


def test_headline_drops_rows_rather_than_cross_the_limit():
    full = _round5_record()
    full["workloads"] = full["workloads"] * 8
    for index, entry in enumerate(full["workloads"]):
        full["workloads"][index] = dict(entry, workload=f"{entry['workload']} #{index}")
    line = bench.compact_headline(full)
    assert len(json.dumps(line)) < 4096 and line["workloads_dropped_for_size"] > 0 and "roofline" in line and "cpu_baseline" in line
