"""Data-parallel gradient exchange: one process per GPU, RCCL (torch.distributed backend "nccl") over xGMI.

Replaces what ``nn.DataParallel`` does implicitly at reference ``train.py:190-192,278-280`` (GPU0-rooted
scatter, per-step 326 MB weight broadcast, reduce-add of all gradients to GPU0): every rank keeps its own
replica and its own shard of the batch; the only exchange per step is ONE all-reduce (mean) of the flat
81.57 M-element gradient buffer the engine's backward pass fills -- issued as a few large buckets so RCCL
can pipeline them over the 7 xGMI links while the later buckets are still queued.

BatchNorm statistics stay per replica, exactly like the reference's DataParallel (no SyncBN).
"""
import torch

DEFAULT_BUCKET_BYTES = 64 << 20
FORCE_COLLECTIVES = False      # True: issue the collectives also in a single-rank group (exercises the RCCL path on one GPU)


def allreduce_sum_async(flat, group=None, bucket_bytes=DEFAULT_BUCKET_BYTES):
    """Start the in-place sum over the ranks of `group` of a 1-D contiguous tensor, in buckets; returns the work
    handles ([] when there is nothing to exchange).  With the nccl (= RCCL) backend the collectives run on RCCL's own
    stream behind everything enqueued on the current stream so far -- kernels enqueued afterwards overlap with them."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size(group) == 1 and not FORCE_COLLECTIVES):
        return []
    assert flat.dim() == 1 and flat.is_contiguous()
    n = flat.numel()
    per = max(1, bucket_bytes // flat.element_size())
    return [dist.all_reduce(flat[lo:min(n, lo + per)], op=dist.ReduceOp.SUM, group=group, async_op=True)
            for lo in range(0, n, per)]


def allreduce_mean_(flat, group=None, bucket_bytes=DEFAULT_BUCKET_BYTES):
    """In-place mean over the ranks of `group` of a 1-D contiguous tensor, in buckets.  No-op when
    torch.distributed is not initialised or the group has a single rank."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return flat
    world = dist.get_world_size(group)
    if world == 1:
        return flat
    assert flat.dim() == 1 and flat.is_contiguous()
    n = flat.numel()
    per = max(1, bucket_bytes // flat.element_size())
    works = []
    for lo in range(0, n, per):
        works.append(dist.all_reduce(flat[lo:min(n, lo + per)], op=dist.ReduceOp.SUM, group=group, async_op=True))
    for w in works:
        w.wait()
    flat.mul_(1.0 / world)
    return flat


def broadcast_module_(module, src=0, group=None):
    """Make every rank start from rank `src`'s parameters and buffers (what DataParallel's per-step
    replicate() did implicitly; here it is needed once, after construction / checkpoint load)."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return module
    with torch.no_grad():
        for t in list(module.parameters()) + list(module.buffers()):
            dist.broadcast(t, src=src, group=group)
    return module
