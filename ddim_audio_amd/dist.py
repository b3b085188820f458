"""Batch-parallel sampling across the GPUs of one node: one process per GPU, no collective in the
step loop (every op of the path is per-sample: SURVEY section 8e), optional final all-gather over
RCCL (``backend="nccl"`` on ROCm) or gloo.  Training is data parallel: identical replicas, one all-reduce of the flat
gradient buffer per step (``attach_grad_sync``)."""
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


def make_grad_sync(group=None, bucket_mb=64):
    """Data-parallel gradient averaging for the training step: the backward writes every parameter
    gradient into ONE flat fp32 buffer (47.2 M elements = 188.6 MB for configs/audio.yml), which is all-reduced here in
    ``bucket_mb`` slices issued back to back (RCCL pipelines them over the xGMI links) and scaled by 1/world.  The local
    loss is a batch mean (functions/losses.py:18), so sum/world of the rank gradients is the global-batch gradient."""
    def sync(flat):
        if not dist.is_initialized():
            return flat
        world = dist.get_world_size(group)
        if world == 1:
            return flat
        n = max(1, (bucket_mb << 20) // flat.element_size())
        works = [dist.all_reduce(flat[i:i + n], op=dist.ReduceOp.SUM, group=group, async_op=True) for i in range(0, flat.numel(), n)]
        for w in works:
            w.wait()
        flat.mul_(1.0 / world)
        return flat
    return sync


def attach_grad_sync(model, group=None, bucket_mb=64):
    """Make ``model``'s backward average its gradients over the ranks of ``group`` (one process per GPU)."""
    model.grad_sync = make_grad_sync(group, bucket_mb)
    return model
