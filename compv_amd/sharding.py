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


def max_over_ranks(value, dist=None, device=None):
    """MAX of a python float over all ranks (the timing rule of bench.py); identity when not distributed."""
    if dist is None or not dist.is_initialized():   # a world of ONE initialised rank still goes through the collective (bench.py --force-dist)
        return float(value)
    import torch
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


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
