#!/usr/bin/env python3
"""Measurement aid (round 5): what would XCD-affinity-only binning buy the 256-clip batch? The instances are
permuted on the HOST so that slot s -- workgroup s / 4, which runs on XCD (s / 4) % 8 -- holds an instance whose clip % 8 is that XCD
(each L2 then sees 32 of the 256 clips), in no other order; the decode of that list is timed against the list as drawn and in full
locality order. If the affine order does not clearly beat the list as drawn by more than a binning pass costs, the device side
binning kernel need not be built."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    import torch
    job = bench.Job("256_clips", 0, 0)
    n = job.num_instances
    job.prewarm(0.1)
    as_drawn = min(job.kernel_ms(300) for _ in range(3)) * 1000

    clip_of = job.clip_indices.astype(np.int64)
    slots_of_xcd = [np.array([s for s in range(n) if (s // 4) % 8 == x]) for x in range(8)]
    for label, key in (("clip % 8 on its XCD, otherwise as drawn", None), ("clip % 8 on its XCD, bucketed by clip inside the XCD", "clip")):
        permutation = np.full(n, -1, dtype=np.int64)
        leftovers, free = [], []
        for x in range(8):
            mine = np.nonzero(clip_of % 8 == x)[0]
            if key == "clip":
                mine = mine[np.argsort(clip_of[mine], kind="stable")]
            take = min(mine.size, slots_of_xcd[x].size)
            permutation[slots_of_xcd[x][:take]] = mine[:take]
            leftovers += list(mine[take:])
            free += list(slots_of_xcd[x][take:])
        permutation[np.array(free, dtype=np.int64)] = np.array(leftovers, dtype=np.int64)
        assert np.array_equal(np.sort(permutation), np.arange(n))
        misplaced = int(np.sum(clip_of[permutation] % 8 != (np.arange(n) // 4) % 8))
        job.d_clips.copy_(torch.from_numpy(job.handles[job.clip_indices[permutation]].astype(np.int32)))
        job.d_times.copy_(torch.from_numpy(job.times[permutation]))
        torch.cuda.synchronize()
        t = min(job.kernel_ms(300) for _ in range(3)) * 1000
        print(f"{label:60s} {t:7.2f} us   ({misplaced} of {n} instances not on their XCD)")
    print(f"{'as drawn':60s} {as_drawn:7.2f} us")
    job.close()
    ordered = bench.Job("256_clips", 0, 0, order="locality")
    ordered.prewarm(0.1)
    print(f"{'aclhip_order_instances_for_locality order':60s} {min(ordered.kernel_ms(300) for _ in range(3)) * 1000:7.2f} us")
    ordered.close()


if __name__ == "__main__":
    main()
