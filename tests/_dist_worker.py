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

    # ---- the data path of bench.py --scatter: the step's global batch is born on rank 0, every rank receives its block with
    # one grouped send/recv, the per-frame line counts and strongest lines are all-gathered.  Same functions, CPU tensors, gloo.
    import numpy as np
    import torch
    F, H, W, K = 3, 24, 40, 4
    blocks = None
    if rank == 0:
        blocks = [torch.from_numpy(np.stack([synth_frame(W, H, 100 * r + f) for f in range(F)])) for r in range(world)]
    mine_in = torch.zeros((F, H, W), dtype=torch.uint8)
    sharding.scatter_blocks(dist, (lambda r: blocks[r]), mine_in, src=0)
    exp_in = np.stack([synth_frame(W, H, 100 * rank + f) for f in range(F)])
    scatter_ok = bool((mine_in.numpy() == exp_in).all())
    my_counts = torch.tensor([1000 * rank + f for f in range(F)], dtype=torch.int32)
    my_lines = (torch.arange(F * K * 5, dtype=torch.int32).reshape(F, K, 5) + 100000 * rank)
    g_counts, g_lines = sharding.gather_lines(dist, my_counts, my_lines)
    ok = torch.tensor([1 if scatter_ok else 0], dtype=torch.int32)
    dist.all_reduce(ok, op=dist.ReduceOp.MIN)
    seen = sharding.ranks_seen(dist)          # what bench.py puts into its line as n_ranks_seen / ranks_seen
    if rank == 0:
        with open(out_path, "w") as f:
            json.dump({"world": world, "counts": allc, "tmax": tmax, "elapsed_rank0": elapsed, "scatter_ok_all_ranks": int(ok.item()),
                       "gathered_counts": g_counts.tolist(), "gathered_lines_shape": list(g_lines.shape),
                       "gathered_lines_first": [int(g_lines[r * F, 0, 0]) for r in range(world)], "ranks_seen": seen}, f)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
