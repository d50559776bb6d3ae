"""Data parallelism for the training step (new relative to the reference, which is single-GPU:
SURVEY.md section 5 / 8e).

Dialogs are independent in forward and backward, so the step shards by dialog: every rank owns
`batchSize` dialogs, the flat gradient vector (wrapperdW) is summed over RCCL/xGMI
(`torch.distributed` backend "nccl" == RCCL on ROCm) and averaged, and only THEN clamped and fed to
Adam on every rank -- which equals the single-process step on the concatenated batch because the
loss is a mean over rows (model.lua:38).  No other collective exists on the path.
"""
import torch
import torch.distributed as dist


def shard_dialogs(num_dialogs, rank, world):
    """Contiguous split of dialog indices [0, num_dialogs) over ranks (first ranks get the remainder)."""
    q, r = divmod(num_dialogs, world)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def reduce_gradients(flat_grad, group=None, async_op=False):
    """Sum the flat gradient over the group.  Returns (gscale, work): multiply by gscale (= 1/world)
    inside the fused clamp+Adam kernel.  Works for HIP tensors (RCCL) and CPU tensors (gloo tests)."""
    if group is None and not dist.is_initialized():
        return 1.0, None
    world = dist.get_world_size(group)
    work = dist.all_reduce(flat_grad, op=dist.ReduceOp.SUM, group=group, async_op=async_op)
    return 1.0 / world, work


def rank_loss_mean(local_loss, group=None):
    """Global mean of the per-rank mean losses (logging only; off the critical path)."""
    if group is None and not dist.is_initialized():
        return float(local_loss)
    t = torch.tensor([float(local_loss)], dtype=torch.float64)
    if dist.get_backend(group) == 'nccl':
        t = t.cuda()
    dist.all_reduce(t, group=group)
    return float(t.item()) / dist.get_world_size(group)
