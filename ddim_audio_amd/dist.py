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


def make_grad_sync(group=None, bucket_mb=64, overlap=None, min_world=2):
    """Data-parallel gradient averaging for the training step: the backward writes every parameter
    gradient into ONE flat fp32 buffer (47.2 M elements = 188.6 MB for configs/audio.yml).  ``sync(flat)`` averages it over the
    ranks in ``bucket_mb`` slices issued back to back (RCCL pipelines them over the xGMI links).  With ``overlap`` (RCCL /
    ``nccl`` backend) the backward instead hands over three buckets in the order it finishes them -- the up path (34 MB), the
    FNet bottleneck (109 MB), then the down path + timestep MLP (45 MB) -- each behind a HIP event, and ``sync.staged`` issues
    each bucket's all-reduce on a side stream once its event has fired: only the last bucket's collective is exposed.  The
    local loss is a batch mean (functions/losses.py:18), so the rank average of the gradients is the global-batch gradient.
    The 1/world factor rides the collective (``ReduceOp.AVG``) on RCCL -- no extra pass over the 188.6 MB; gloo (CPU tests) has
    no AVG, there it is one in-place multiply.  ``overlap=None``: on, unless ``DDIMX_GRAD_OVERLAP=0``.  ``min_world``: smallest
    world size for which the staged path is taken (2; the one-GPU test box runs it for real with a 1-rank RCCL group)."""
    import os
    if overlap is None:
        overlap = os.environ.get("DDIMX_GRAD_OVERLAP", "1") != "0"
    state = {"side": None}

    def _slices(flat, lo, hi):
        n = max(1, (bucket_mb << 20) // flat.element_size())
        return [flat[i:min(i + n, hi)] for i in range(lo, hi, n)]

    def _reduce(views):
        """Async averaged all-reduce of each view; returns (works, needs_scale)."""
        avg = dist.get_backend(group) == "nccl"
        op = dist.ReduceOp.AVG if avg else dist.ReduceOp.SUM
        return [dist.all_reduce(v, op=op, group=group, async_op=True) for v in views], not avg

    def active():
        return (overlap and dist.is_initialized() and dist.get_world_size(group) >= min_world and dist.get_backend(group) == "nccl")

    def sync(flat):
        if not dist.is_initialized():
            return flat
        world = dist.get_world_size(group)
        if world == 1:
            return flat
        works, scale = _reduce(_slices(flat, 0, flat.numel()))
        for w in works:
            w.wait()
        if scale:
            flat.mul_(1.0 / world)
        return flat

    def staged(flat, ranges, events=None):
        """ranges: [(begin, end)] float offsets per bucket in readiness order; events: the torch.cuda.Event the backward records
        when the bucket is final (None on the CPU / gloo rehearsal path, where the buckets are final already: same bucket
        plumbing, no streams).  Returns after making the CURRENT stream wait for every collective."""
        world = dist.get_world_size(group)
        works, scale = [], False
        if events is None or not flat.is_cuda:
            for lo, hi in ranges:
                if hi > lo:
                    w, scale = _reduce(_slices(flat, lo, hi))
                    works += w
        else:
            if state["side"] is None:
                state["side"] = torch.cuda.Stream(device=flat.device)
            side = state["side"]
            for (lo, hi), ev in zip(ranges, events):
                if hi <= lo:
                    continue
                side.wait_event(ev)
                with torch.cuda.stream(side):  # the collective is ordered behind `side`, i.e. behind the bucket's event
                    w, scale = _reduce(_slices(flat, lo, hi))
                    works += w
        for w in works:
            w.wait()                       # current stream waits for the RCCL stream
        if scale:
            covered = sorted((lo, hi) for lo, hi in ranges if hi > lo)
            for lo, hi in covered:
                flat[lo:hi].mul_(1.0 / world)
        return flat

    sync.staged = staged
    sync.active = active
    return sync


def attach_grad_sync(model, group=None, bucket_mb=64, overlap=None, min_world=2):
    """Make ``model``'s backward average its gradients over the ranks of ``group`` (one process per GPU)."""
    model.grad_sync = make_grad_sync(group, bucket_mb, overlap, min_world)
    return model
