#!/usr/bin/env python3
"""Measurement aid: kernel time of bench.py workloads under several environment variants, one process per (variant, workload).

usage: python tools/variant_sweep.py [--repeats N] [--workloads cinematic,one_clip] label[:ENV=VALUE[,ENV=VALUE...]] ...

Every variant is run in its own process (the library reads its measurement knobs once); prints one line per (variant, workload):
label workload kernel_us frac kernel_name. Results also go to gpurun_out/variant_sweep.jsonl.
"""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

RUNNER = r"""
import json, sys
sys.path.insert(0, %(root)r)
import bench
result = bench.measure_job(%(workload)r, 0, 0, repeats=%(repeats)d, **%(options)r)
print("RESULT " + json.dumps(result))
"""


def main():
    parser = argparse.ArgumentParser()
    parser.add_argument("--repeats", type=int, default=400)
    parser.add_argument("--workloads", default="cinematic")
    parser.add_argument("--order", default=None)
    parser.add_argument("--layout", default=None)
    parser.add_argument("variants", nargs="+")
    args = parser.parse_args()

    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    log = open(os.path.join(ROOT, "gpurun_out", "variant_sweep.jsonl"), "a")
    for variant in args.variants:
        label, _, settings = variant.partition(":")
        env = dict(os.environ)
        for setting in filter(None, settings.split(",")):
            key, _, value = setting.partition("=")
            env[key] = value
        for workload in args.workloads.split(","):
            options = {}
            if args.order:
                options["order"] = args.order
            if args.layout:
                options["layout"] = args.layout
            code = RUNNER % {"root": ROOT, "workload": workload, "repeats": args.repeats, "options": options}
            try:
                proc = subprocess.run([sys.executable, "-c", code], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=240, cwd=ROOT)
            except subprocess.TimeoutExpired:
                print(f"{label:28s} {workload:22s} TIMEOUT", flush=True)
                continue
            line = next((l for l in proc.stdout.splitlines() if l.startswith("RESULT ")), None)
            if line is None:
                print(f"{label:28s} {workload:22s} FAILED rc={proc.returncode} {proc.stderr.strip().splitlines()[-1:] }", flush=True)
                continue
            result = json.loads(line[7:])
            result["variant"] = variant
            log.write(json.dumps(result) + "\n")
            log.flush()
            print(f"{label:28s} {workload:22s} {result['kernel_ms'] * 1000:8.2f} us  frac {result['frac']:.4f}  {result['kernel']}", flush=True)


if __name__ == "__main__":
    main()
