#!/usr/bin/env python3
"""Prints the "round on one screen" markdown table of DESIGN.md 6 from a default run's full record (bench_details.json /
profiles/rNN_bench_details.json). usage: round_table.py <bench_details.json>"""
import json
import sys


def main():
    d = json.load(open(sys.argv[1]))
    r = d["roofline"]
    print("| workload | kernel | us | of 8 TB/s | bound: floor us -> of that bound | HBM traffic / algorithmic MB |")
    print("|---|---|---|---|---|---|")
    head_floor = max(r.get("hbm_floor_ms") or 0.0, r.get("valu_issue_floor_ms") or 0.0) * 1e3
    print(f"| **headline: {d['config']['workload'][:60]}** | `{r['kernel']}` | **{r['kernel_ms'] * 1e3:.1f}** | **{r['frac']:.3f}** | hbm {r['hbm_floor_ms'] * 1e3:.1f} (valu {r['valu_issue_floor_ms'] * 1e3:.1f}) -> {r['frac_of_bound']:.2f} | {r['traffic'] / 1e6:.1f} / {r['algorithmic_bytes_per_launch'] / 1e6:.1f} |")
    for e in d["workloads"]:
        name = e["workload"]
        if e.get("order", "random") != "random" and "order" not in name:
            name += f", {e['order']} order"
        if e.get("layout", "qvv48") != "qvv48" and e["layout"] not in name:
            name += f", {e['layout']}"
        floors = ""
        if e.get("bound"):
            floor = (e.get("valu_issue_floor_ms") if e["bound"] == "valu" else e.get("hbm_floor_ms")) or 0.0
            floors = f"{e['bound']} {floor * 1e3:.1f} -> {e['frac_of_bound']:.2f}"
        traffic = f"{e['traffic'] / 1e6:.1f} / {e['algorithmic_bytes'] / 1e6:.1f}" if e.get("traffic") else "-"
        print(f"| {name}{' (fast)' if e.get('fast') else ''} | `{e['kernel']}` | {e['kernel_ms'] * 1e3:.1f} | {e['frac']:.3f} | {floors} | {traffic} |")
    c = d["cpu_baseline"]
    print(f"\nCPU baseline ({c['kind']}): {c['value'] / 1e6:.1f} M poses/s at {c['threads_at_best']} threads (cgroup {c.get('cgroup_cpu_max')}, nproc {c['nproc']}, {c['physical_cores']} physical cores), "
          f"{c['per_thread_1t'] / 1e6:.2f} M per thread; GPU / CPU {c['gpu_over_cpu']:.1f} x the quota, {c['gpu_over_cpu_extrapolated_physical_cores']:.1f} x extrapolated to the physical cores.")
    print(f"self_check: {d['self_check']}")


if __name__ == "__main__":
    main()
