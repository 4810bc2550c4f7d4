"""One process per GPU (torch.distributed; backend "nccl" = RCCL on ROCm, "gloo" on CPU for the tests).

The massive-body path does not shard: ranks integrate independent replicas and only the timing is
reduced (max over ranks) -- no data-path collective. The massless sweep (next round) shards spacecraft with
`shard_range` and broadcasts the ephemeris table once.
"""
import os


def env_rank():
    """(rank, local_rank, world_size) from the torch.distributed.run environment; (0, 0, 1) when run directly."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def shard_range(n_items, rank, world):
    """Contiguous block partition of n_items independent work items; blocks differ by at most one item."""
    base, extra = divmod(n_items, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def replica_seed(base_seed, rank):
    return base_seed + rank


def reduce_timing(elapsed_s, units_local, dist=None, device="cpu"):
    """Whole-job throughput = (sum of units over ranks) / (max elapsed over ranks). `dist` is torch.distributed
    (initialised) or None for a single process. Returns (total_units, max_elapsed)."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return float(units_local), float(elapsed_s)
    import torch
    t = torch.tensor([float(elapsed_s)], dtype=torch.float64, device=device)
    u = torch.tensor([float(units_local)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.all_reduce(u, op=dist.ReduceOp.SUM)
    return float(u.item()), float(t.item())
