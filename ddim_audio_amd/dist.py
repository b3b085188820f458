"""Batch-parallel sampling across the GPUs of one node: one process per GPU, no collective in the
step loop (every op of the path is per-sample: SURVEY section 8e), optional final all-gather over
RCCL (``backend="nccl"`` on ROCm) or gloo."""
import torch
import torch.distributed as dist


def shard_bounds(n, rank, world):
    """Contiguous, balanced [lo, hi) of n samples for this rank (first n % world ranks get one more)."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_batch(x, rank=None, world=None):
    if rank is None:
        rank, world = (dist.get_rank(), dist.get_world_size()) if dist.is_initialized() else (0, 1)
    lo, hi = shard_bounds(x.size(0), rank, world)
    return x[lo:hi]


def gather_batch(local, n_total, group=None):
    """All-gather ragged per-rank shards back into the full batch (same order as ``shard_batch``)."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return local
    world = dist.get_world_size(group)
    sizes = [shard_bounds(n_total, r, world) for r in range(world)]
    mx = max(hi - lo for lo, hi in sizes)
    pad = torch.zeros((mx,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: local.size(0)] = local
    outs = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(outs, pad, group=group)
    return torch.cat([o[: hi - lo] for o, (lo, hi) in zip(outs, sizes)], dim=0)


def sample_sharded(x_full, sampler, gather=True, group=None):
    """Run ``sampler(x_shard) -> tensor`` on this rank's shard of ``x_full`` (every rank holds the same
    full noise tensor, so results do not depend on the GPU count) and optionally gather the results."""
    n = x_full.size(0)
    local = sampler(shard_batch(x_full).clone())
    return gather_batch(local, n, group) if gather else local
