"""Frame sharding for multi-GPU runs.  Frames are independent units: rank r of `world` owns a contiguous
block of the global batch, no data-path collective is needed; the only communication is a barrier around the timed
region, a MAX-reduction of the elapsed time and (optionally) a gather of tiny per-frame results."""

BASE_SEED = 12345  # frame f of the global synthetic batch uses seed BASE_SEED + f


def shard_range(global_frames, world, rank):
    """Contiguous block partition; the first (global_frames % world) ranks take one extra frame."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError("bad world/rank")
    q, r = divmod(global_frames, world)
    start = rank * q + min(rank, r)
    return range(start, start + q + (1 if rank < r else 0))


def frame_seed(frame_index):
    return BASE_SEED + frame_index


# ---- the benchmark's block rotation (BASELINE config 4: 256 frames = 8 resident blocks of 32) --------------------------------------
def block_seeds(frames_per_block, n_blocks, golden_frames=256):
    """Seeds of the resident blocks every rank builds: frame f of block b is frame (b * F + f) mod golden_frames of the global synthetic batch."""
    return [[frame_seed((b * frames_per_block + f) % golden_frames) for f in range(frames_per_block)] for b in range(n_blocks)]


def block_of_step(rank, step, n_blocks):
    """The resident block rank `rank` processes at step `step`: at any step the ranks of a world of n_blocks hold n_blocks DIFFERENT blocks (with
    8 ranks and 8 blocks of 32: exactly config 4's 256 frames per step), and consecutive steps of a rank never re-read the same block."""
    return (rank + step) % n_blocks


def max_over_ranks(value, dist=None, device=None):
    """MAX of a python float over all ranks (the timing rule of bench.py); identity when not distributed."""
    if dist is None or not dist.is_initialized():   # a world of ONE initialised rank still goes through the collective (bench.py --force-dist)
        return float(value)
    import torch
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def ranks_seen(dist=None, device=None):
    """Every rank contributes its own rank number to an all-gather; the sorted list must be 0 .. world-1 (a SCALE record can be checked against it)."""
    if dist is None or not dist.is_initialized():
        return [0]
    import torch
    world = dist.get_world_size()
    mine = torch.tensor([dist.get_rank()], dtype=torch.int64, device=device)
    out = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(out, mine)
    return sorted(int(t.item()) for t in out)


def gather_frame_results(local_values, dist=None, device=None):
    """All-gather a short list of per-frame integers (e.g. line counts); returns the global list in frame order."""
    if dist is None or not dist.is_initialized():
        return list(local_values)
    import torch
    world = dist.get_world_size()
    n = torch.tensor([len(local_values)], dtype=torch.int64, device=device)
    sizes = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(sizes, n)
    m = int(max(int(s.item()) for s in sizes))
    buf = torch.zeros(m, dtype=torch.int64, device=device)
    buf[:len(local_values)] = torch.tensor(list(local_values), dtype=torch.int64, device=device)
    out = [torch.zeros_like(buf) for _ in range(world)]
    dist.all_gather(out, buf)
    res = []
    for s, o in zip(sizes, out):
        res.extend(int(v) for v in o[:int(s.item())].tolist())
    return res


# ---- optional data path of SURVEY.md 8(e): the step's batch is born on one rank ---------------------------------------------------
def scatter_blocks(dist, block_of_rank, out, src=0):
    """Rank `src` owns the global batch of a step as per-rank blocks (block_of_rank(r) -> tensor shaped like `out`, only called on `src`);
    every rank ends up with its block in `out`.  One grouped send/recv (torch.distributed.batch_isend_irecv: a single RCCL group on
    GPUs, so the P2P copies of all destinations run concurrently over their own xGMI links); no collective involves the payload twice.
    Without an initialised process group it is a plain copy."""
    if dist is None or not dist.is_initialized():
        out.copy_(block_of_rank(0))
        return
    rank, world = dist.get_rank(), dist.get_world_size()
    staged = out.is_cuda and dist.get_backend() == "gloo"   # gloo has no device-to-device send/recv: host staging (test configurations only)
    if rank == src:
        ops = [dist.P2POp(dist.isend, block_of_rank(r).cpu() if staged else block_of_rank(r), r) for r in range(world) if r != src]
        out.copy_(block_of_rank(src))
        tmp = None
    else:
        tmp = out.cpu() if staged else out
        ops = [dist.P2POp(dist.irecv, tmp, src)]
    if ops:
        for w in dist.batch_isend_irecv(ops):
            w.wait()
    if staged and tmp is not None:
        out.copy_(tmp)


def gather_lines(dist, counts, lines_topk):
    """All-gather of the per-frame line counts [F] and of the K strongest lines of every frame [F, K, 5] (int32 rows of compvhip_line):
    every rank ends up with the results of the global batch in rank order.  Returns (counts [world*F], lines [world*F, K, 5])."""
    if dist is None or not dist.is_initialized():
        return counts, lines_topk
    import torch
    world = dist.get_world_size()
    if counts.is_cuda and dist.get_backend() == "gloo":     # host staging (test configurations only)
        c, l = gather_lines(dist, counts.cpu(), lines_topk.cpu())
        return c.to(counts.device), l.to(counts.device)
    all_counts = torch.empty((world * counts.shape[0],), dtype=counts.dtype, device=counts.device)
    all_lines = torch.empty((world * lines_topk.shape[0],) + tuple(lines_topk.shape[1:]), dtype=lines_topk.dtype, device=lines_topk.device)
    dist.all_gather_into_tensor(all_counts, counts.contiguous())
    dist.all_gather_into_tensor(all_lines, lines_topk.contiguous())
    return all_counts, all_lines
