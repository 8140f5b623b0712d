#!/usr/bin/env python3
"""Summarises rocprofv3 --pmc counter_collection CSVs: mean per-dispatch value of every counter for kernels matching a substring."""
import csv
import sys
from collections import defaultdict

def main():
    match = sys.argv[1]
    for path in sys.argv[2:]:
        sums, counts = defaultdict(float), defaultdict(int)
        for row in csv.DictReader(open(path)):
            if match in row["Kernel_Name"]:
                sums[row["Counter_Name"]] += float(row["Counter_Value"])
                counts[row["Counter_Name"]] += 1
        for name in sorted(sums):
            print(f"{path}: {name:28s} mean/dispatch {sums[name] / counts[name]:16.1f}  ({counts[name]} dispatches)")

if __name__ == "__main__":
    main()
