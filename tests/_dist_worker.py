"""Worker for tests/test_distributed.py: launched by torch.distributed.run with the gloo backend (CPU).  It runs
bench.py's multi-rank plumbing (shard_range / barrier / MAX-over-ranks / gather) with the CPU ORACLE standing in for
the per-frame HIP step (test infrastructure only -- the product path never does this)."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import torch.distributed as dist  # noqa: E402

from compv_amd import sharding  # noqa: E402
from oracle_bindings import Oracle, synth_frame  # noqa: E402


def main():
    out_path = sys.argv[1]
    global_frames = int(sys.argv[2])
    dist.init_process_group(backend="gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    orc = Oracle()
    mine = sharding.shard_range(global_frames, world, rank)
    dist.barrier()
    t0 = time.perf_counter()
    counts = []
    for f in mine:
        img = synth_frame(160, 120, sharding.frame_seed(f))
        rc, e = orc.canny(img, 59.0, 119.0)
        counts.append(int((e != 0).sum()))
    time.sleep(0.05 * (rank + 1))   # uneven ranks: MAX must pick the slowest
    dist.barrier()
    elapsed = time.perf_counter() - t0
    tmax = sharding.max_over_ranks(elapsed, dist)
    allc = sharding.gather_frame_results(counts, dist)
    if rank == 0:
        with open(out_path, "w") as f:
            json.dump({"world": world, "counts": allc, "tmax": tmax, "elapsed_rank0": elapsed}, f)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
