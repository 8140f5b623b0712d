#!/usr/bin/env python3
"""Several processes x several streams call aclhip_order_instances_device at the same time, over and over: the one launch form's
workgroups wait for one another at barriers in global memory, and every launch must get all of its workgroups resident.
usage: order_stress.py [processes] [streams per process] [calls per stream]"""
import multiprocessing as mp
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def worker(rank, num_streams, calls, queue, start):
    import numpy as np
    import torch
    from acl_amd import runtime, synth
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
    from test_order_instances import check_order

    clip = synth.build_clip(seed=5, num_tracks=3, num_samples=2)
    device = torch.device("cuda", 0)
    n = 65536
    with runtime.Context(0) as context:
        handles = np.array([context.register_clip(clip.blob, check_hash=False) for _ in range(256)], dtype=np.uint32)
        rng = np.random.default_rng(rank)
        streams = [torch.cuda.Stream(device) for _ in range(num_streams)]
        lists = [handles[rng.integers(0, handles.size, size=n)] for _ in streams]
        buffers = []
        for instance_clips in lists:
            d_clips = torch.from_numpy(instance_clips.astype(np.int32)).to(device)
            d_times = torch.zeros((n,), dtype=torch.float32, device=device)
            d_order = torch.full((n,), -1, dtype=torch.int32, device=device)
            d_out_clips = torch.full((n,), -1, dtype=torch.int32, device=device)
            d_out_times = torch.zeros((n,), dtype=torch.float32, device=device)
            buffers.append((d_clips, d_times, d_order, d_out_clips, d_out_times))
        torch.cuda.synchronize(device)
        start.wait()        # everybody orders at the same time
        t0 = time.perf_counter()
        for call in range(calls):
            for k, stream in enumerate(streams):
                d_clips, d_times, d_order, d_out_clips, d_out_times = buffers[k]
                context.order_instances_device(d_clips.data_ptr(), d_times.data_ptr(), n, d_order.data_ptr(), d_out_clips.data_ptr(), d_out_times.data_ptr(), stream=stream.cuda_stream)
        torch.cuda.synchronize(device)
        elapsed = time.perf_counter() - t0
        for k in range(num_streams):
            order = buffers[k][2].cpu().numpy().astype(np.uint32)
            check_order(lists[k], order, 1, stable=False)
            assert np.array_equal(buffers[k][3].cpu().numpy().astype(np.uint32), lists[k][order])
    queue.put((rank, elapsed))


def main():
    processes = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    num_streams = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    calls = int(sys.argv[3]) if len(sys.argv) > 3 else 2000
    mp.set_start_method("spawn")
    queue = mp.Queue()
    start = mp.Barrier(processes)
    workers = [mp.Process(target=worker, args=(rank, num_streams, calls, queue, start)) for rank in range(processes)]
    for w in workers:
        w.start()
    for w in workers:
        w.join(600)
    codes = [w.exitcode for w in workers]
    results = []
    while not queue.empty():
        results.append(queue.get())
    print("exit codes", codes)
    for rank, elapsed in sorted(results):
        print(f"  process {rank}: {calls * num_streams} orderings in {elapsed:.2f} s = {elapsed / (calls * num_streams) * 1e6:.1f} us each")
    if any(code != 0 for code in codes) or len(results) != processes:
        raise SystemExit("FAILED")
    print("ok")


if __name__ == "__main__":
    main()
