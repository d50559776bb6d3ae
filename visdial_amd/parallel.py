"""Data parallelism for the training step (new relative to the reference, which is single-GPU:
SURVEY.md section 5 / 8e).

Dialogs are independent in forward and backward, so the step shards by dialog: every rank owns
`batchSize` dialogs, the flat gradient vector (wrapperdW) is summed over RCCL/xGMI
(`torch.distributed` backend "nccl" == RCCL on ROCm) and averaged, and only THEN clamped and fed to
Adam on every rank -- which equals the single-process step on the concatenated batch because the
loss is a mean over rows (model.lua:38).  No other collective exists on the path.
"""
import ctypes as C

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


# ---------------------------------------------------------------------------------------------------------------
# The collective behind the C ABI (csrc/comm.hip): the library owns the RCCL communicator and its stream; the host only
# carries the 128-byte rendezvous token from rank 0 to the peers.  This is the path a Lua host uses too
# (lua/model.lua:initComm); torch.distributed below is nothing but the courier of those 128 bytes.
def library_comm_world():
    """world size of the library's communicator; 0 = none"""
    from . import _lib
    r, w = C.c_int(), C.c_int()
    _lib.call("vd_comm_info", C.byref(r), C.byref(w))
    return int(w.value)


def init_library_comm(rank, world, exchange):
    """Join the library's RCCL communicator on the current device.  `exchange(token_or_None) -> token`: called with
    the 128-byte token on rank 0 and None elsewhere, returns rank 0's token on every rank (any channel will do)."""
    from . import _lib
    buf = C.create_string_buffer(128)
    if rank == 0:
        _lib.call("vd_comm_unique_id", buf)
    token = exchange(bytes(buf.raw) if rank == 0 else None)
    assert isinstance(token, (bytes, bytearray)) and len(token) == 128
    _lib.call("vd_comm_init", int(rank), int(world), C.create_string_buffer(bytes(token), 128))


def init_library_comm_over(group=None):
    """the same with a torch.distributed group (any backend, gloo is enough) as the courier"""
    rank, world = dist.get_rank(group), dist.get_world_size(group)

    def exchange(token):
        box = [token]
        dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        return box[0]
    init_library_comm(rank, world, exchange)


def destroy_library_comm():
    from . import _lib
    _lib.call("vd_comm_destroy")


def library_comm_available():
    """(ok, NCCL_VERSION_CODE, error text): can THIS process load RCCL (no collective, no device needed)"""
    from . import _lib
    v = C.c_int(0)
    rc = _lib.load().vd_comm_available(C.byref(v))
    return rc == 0, int(v.value), ('' if rc == 0 else _lib.load().vd_last_error().decode('utf-8', 'replace'))


def library_comm_stats():
    """the last vd_model_allreduce_grads of this process: dict(bucket1_floats, bucket2_floats, overlapped, calls)"""
    from . import _lib
    b1, b2, ov, n = C.c_int64(), C.c_int64(), C.c_int(), C.c_int64()
    _lib.call("vd_comm_stats", C.byref(b1), C.byref(b2), C.byref(ov), C.byref(n))
    out = dict(bucket1_floats=int(b1.value), bucket2_floats=int(b2.value), overlapped=bool(ov.value), calls=int(n.value))
    if out['overlapped']:
        # how long before the end of the backward pass the encoder bucket was already summed (negative: exposed)
        lead = C.c_float()
        if _lib.load().vd_comm_overlap_ms(C.byref(lead)) == 0:
            out['bucket1_done_before_backward_end_ms'] = round(float(lead.value), 3)
    return out


def join_library_comm(group, device=None):
    """Every rank of `group` (any backend; gloo is enough -- it only carries flags and the 128-byte token) joins the
    library's RCCL communicator, or none does.  Three agreed phases, so that no rank can strand its peers inside a collective:
      1. each rank checks that it can load RCCL (vd_comm_available: no collective)      -> MIN over ranks, stop if 0
      2. rank 0 makes the token (vd_comm_unique_id)                                      -> broadcast (None = failed), stop if None
      3. vd_comm_init on every rank (collective inside RCCL)                            -> MIN over ranks; a rank that joined
         while another failed destroys its communicator again
    Returns (ok, report): report = one line per phase for this rank's log."""
    from . import _lib
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    agree = lambda ok: bool(_min_flag(ok, group))
    report = []
    if device is not None:
        try:
            _lib.call("vd_set_device", int(device))
        except Exception as exc:
            report.append('vd_set_device(%s) failed: %s' % (device, exc))
    ok, version, err = library_comm_available()
    report.append('librccl: %s' % ('NCCL_VERSION_CODE %d' % version if ok else 'NOT loadable (%s)' % err))
    if not agree(ok and not any('vd_set_device' in r for r in report)):
        report.append('a rank cannot load RCCL or select its device: no rank joins')
        return False, report
    token = None
    if rank == 0:
        buf = C.create_string_buffer(128)
        if _lib.load().vd_comm_unique_id(buf) == 0:
            token = bytes(buf.raw)
        else:
            report.append('vd_comm_unique_id failed: %s' % _lib.load().vd_last_error().decode('utf-8', 'replace'))
    box = [token]
    dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
    if box[0] is None:
        report.append('rank 0 could not create the rendezvous token: no rank joins')
        return False, report
    rc = _lib.load().vd_comm_init(int(rank), int(world), C.create_string_buffer(bytes(box[0]), 128))
    if rc != 0:
        report.append('vd_comm_init failed: %s' % _lib.load().vd_last_error().decode('utf-8', 'replace'))
    if not agree(rc == 0):
        if rc == 0:
            destroy_library_comm()
        report.append('a rank failed to join: communicator abandoned on every rank')
        return False, report
    report.append('joined: rank %d of %d' % (rank, world))
    return True, report


def _min_flag(ok, group):
    flag = torch.tensor([1 if ok else 0], dtype=torch.int32)
    if dist.get_backend(group) == 'nccl':
        flag = flag.cuda()
    dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
    return int(flag.item())
