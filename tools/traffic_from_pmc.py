#!/usr/bin/env python3
"""Turns two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs) into the per-launch HBM traffic of the decode kernel.
Units: both counters are KiB; on gfx950 FETCH_SIZE reports half of the bytes of wide coalesced reads (MI355X_MICROARCH.md, HBM section)
and is doubled here. Usage: traffic_from_pmc.py <workload> <fetch.csv> <write.csv> [out.json]  (appends/updates profiles/traffic.json)"""
import csv
import json
import os
import sys


def mean_counter(path, counter, match="decompress_"):
    values, name = [], None
    for row in csv.DictReader(open(path)):
        if row["Counter_Name"] == counter and match in row["Kernel_Name"]:
            values.append(float(row["Counter_Value"]))
            name = row["Kernel_Name"].split("(")[0].split("<")[0].split("::")[-1]
    return sum(values) / len(values), len(values), name


def main():
    workload, fetch_path, write_path = sys.argv[1:4]
    out_path = sys.argv[4] if len(sys.argv) > 4 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "traffic.json")
    fetch_kib, fetch_n, kernel = mean_counter(fetch_path, "FETCH_SIZE")
    write_kib, write_n, _ = mean_counter(write_path, "WRITE_SIZE")
    entry = {
        "workload": workload, "kernel": kernel,
        "fetch_size_kib_mean": fetch_kib, "write_size_kib_mean": write_kib, "dispatches": [fetch_n, write_n],
        "traffic_bytes_per_launch": int((2.0 * fetch_kib + write_kib) * 1024.0),
        "note": "FETCH_SIZE doubled (gfx950 reports half of wide coalesced reads); WRITE_SIZE as reported",
        "sources": [os.path.basename(fetch_path), os.path.basename(write_path)],
    }
    entries = []
    if os.path.exists(out_path):
        entries = [e for e in json.load(open(out_path)) if not (e["workload"] == workload and e["kernel"] == kernel)]
    entries.append(entry)
    json.dump(entries, open(out_path, "w"), indent=1)
    print(entry)


if __name__ == "__main__":
    main()
