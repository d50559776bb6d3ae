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


class _DoneWork(object):
    def wait(self):
        return True


def reduce_gradients(flat_grad, group=None, async_op=False):
    """Sum the flat gradient over the group.  Returns (gscale, work): multiply by gscale (= 1/world)
    inside the fused clamp+Adam kernel.  HIP tensors go over RCCL (backend "nccl"); CPU tensors over gloo.
    A HIP tensor on a gloo group (the 2-ranks-on-one-GPU test, where RCCL refuses duplicate devices) is staged
    through host memory -- a test vehicle only, never the production path."""
    if group is None and not dist.is_initialized():
        return 1.0, None
    world = dist.get_world_size(group)
    if flat_grad.is_cuda and dist.get_backend(group) == 'gloo':
        host = flat_grad.detach().cpu()             # synchronises the current stream
        dist.all_reduce(host, op=dist.ReduceOp.SUM, group=group)
        flat_grad.copy_(host)
        return 1.0 / world, (_DoneWork() if async_op else None)
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
