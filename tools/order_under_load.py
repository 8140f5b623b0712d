#!/usr/bin/env python3
"""Measurement aid: aclhip_order_instances_device on two streams and from a replayed hipGraph WHILE other processes keep the device busy
(start them first: `python bench.py --workload cinematic --steps 400000 --no-extras --no-cpu-baseline &` x 3). Prints what every phase
saw -- valid permutation? bucketed by clip? identity (the give-up's order)? an error from the library? -- so that a run that dies says where.
usage: order_under_load.py [rounds] [replays]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from acl_amd import runtime, synth  # noqa: E402
from test_order_instances import check_order  # noqa: E402


def verdict(clips, order):
    n = clips.size
    if not np.array_equal(np.sort(order), np.arange(n)):
        bad = int(np.sum(order >= n))
        return f"NOT A PERMUTATION ({bad} entries out of range, {np.unique(order).size} distinct)"
    if np.array_equal(order, np.arange(n)):
        return "identity (a give-up)"
    try:
        check_order(clips, order, 1, stable=False)
        return "ok"
    except AssertionError:
        return "permutation, NOT bucketed"


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    replays = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    device = torch.device("cuda", 0)
    clip = synth.build_clip(seed=5, num_tracks=3, num_samples=2)
    t_start = time.time()

    def say(*args):
        print(f"[{time.time() - t_start:6.2f}s]", *args, flush=True)

    with runtime.Context(0) as context:
        handles = np.array([context.register_clip(clip.blob, check_hash=False) for _ in range(200)], dtype=np.uint32)
        rng = np.random.default_rng(9)
        n = 65536
        streams = [torch.cuda.Stream(device) for _ in range(2)]
        buffers = []
        for _ in streams:
            buffers.append((torch.zeros((n,), dtype=torch.int32, device=device), torch.zeros((n,), dtype=torch.float32, device=device),
                            torch.full((n,), -1, dtype=torch.int32, device=device), torch.full((n,), -1, dtype=torch.int32, device=device),
                            torch.zeros((n,), dtype=torch.float32, device=device)))
        torch.cuda.synchronize(device)

        def call(k):
            d_clips, d_times, d_order, d_out_clips, d_out_times = buffers[k]
            try:
                context.order_instances_device(d_clips.data_ptr(), d_times.data_ptr(), n, d_order.data_ptr(), d_out_clips.data_ptr(), d_out_times.data_ptr(), stream=streams[k].cuda_stream)
                return None
            except runtime.AclHipError as error:
                return str(error)[:120]

        for k in range(2):
            call(k)
        torch.cuda.synchronize(device)
        say("first calls done")
        for round_index in range(rounds):
            lists = [handles[rng.integers(0, handles.size, size=n)] for _ in range(2)]
            for k in range(2):
                buffers[k][0].copy_(torch.from_numpy(lists[k].astype(np.int32)))
                buffers[k][2].fill_(-1)
            torch.cuda.synchronize(device)
            errors = []
            for _ in range(8):
                for k in range(2):
                    error = call(k)
                    if error:
                        errors.append((k, error))
            torch.cuda.synchronize(device)
            for k in range(2):
                order = buffers[k][2].cpu().numpy().astype(np.int64)
                say(f"round {round_index} stream {k}:", verdict(lists[k], order), "errors:", [e for e in errors if e[0] == k][:1])

        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=streams[0]):
            error = call(0)
        say("captured", error)
        for replay in range(replays):
            instance_clips = handles[rng.integers(0, handles.size, size=n)]
            buffers[0][0].copy_(torch.from_numpy(instance_clips.astype(np.int32)))
            buffers[0][2].fill_(-1)
            torch.cuda.synchronize(device)
            say(f"replay {replay}: launching")
            if os.environ.get("ORDER_REPLAY_ON_CURRENT_STREAM", "0") == "1":
                graph.replay()          # (the default stream: next to the plain call below, on one scratch -- what the library's guard is for)
            else:
                with torch.cuda.stream(streams[0]):
                    graph.replay()
            error = call(0) if replay == 2 else None
            torch.cuda.synchronize(device)
            order = buffers[0][2].cpu().numpy().astype(np.int64)
            say(f"replay {replay}:", verdict(instance_clips, order), "error:", error)
        del graph
    say("done")


if __name__ == "__main__":
    main()
