"""Worker of tests/test_dp_gpu.py (launched by torch.distributed.run, one process per rank, all on cuda:0).

Every rank builds the HIP `Model` with the process group, trains ONE step on its shard of a global batch of
`world * per_rank` dialogs (two-bucket gradient all-reduce: the encoder bucket is launched asynchronously from the
encoder side stream, the rest at the end of the step -- Model.update / Model._forwardBackward_disc), and rank 0
checks the result against a single-process HIP model stepping on the concatenated batch: the averaged gradient and
the post-Adam parameters must agree (the data-parallel step IS the big-batch step)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

import numpy as np
import torch
import torch.distributed as dist

from conftest import small_params
from visdial_amd.dataloader import SyntheticDataloader
from visdial_amd.model import Model
from visdial_amd.opts import derive
from visdial_amd.parallel import shard_dialogs


def main():
    backend = os.environ.get('VD_TEST_BACKEND', 'nccl')
    rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
    # VD_TEST_DISTINCT_DEVICES=1 (boxes with >= world GPUs): rank r drives cuda:r -- the collective meets a real peer over xGMI
    dev = int(os.environ.get('LOCAL_RANK', 0)) if os.environ.get('VD_TEST_DISTINCT_DEVICES') == '1' else 0
    torch.cuda.set_device(dev)
    if backend == 'nccl':
        dist.init_process_group('nccl', device_id=torch.device('cuda', dev))
    else:
        dist.init_process_group('gloo')
    per_rank = 2
    p = derive(small_params(batchSize=per_rank, gpuid=dev, rank=0))     # same seed on every rank: replicated parameters
    full = SyntheticDataloader(derive(small_params(batchSize=per_rank * world)), seed=123).getTrainBatch(
        derive(small_params(batchSize=per_rank * world)))
    lo, hi = shard_dialogs(per_rank * world, rank, world)
    R = p['maxQuesCount']
    mine = {'ques_fwd': full['ques_fwd'][lo:hi], 'hist': full['hist'][lo:hi], 'img_feat': full['img_feat'][lo:hi],
            'options': full['options'][lo * R:hi * R], 'answer_ind': full['answer_ind'][lo * R:hi * R]}
    host = os.environ.get('VD_TEST_HOST', 'python')
    native = host in ('native', 'native-lib')
    flat = lambda d, names: np.concatenate([np.asarray(d[k], np.float32).reshape(-1) for k in names])
    if native:
        from visdial_amd.native import NativeModel
        if host == 'native-lib':
            # the collective behind the C ABI (csrc/comm.hip): the library owns the RCCL communicator and its stream;
            # the process group (gloo) only carries the 128-byte rendezvous token -- what a Lua host would do by file
            from visdial_amd import _lib
            from visdial_amd.parallel import init_library_comm_over, library_comm_world
            _lib.call("vd_set_device", dev)
            init_library_comm_over(dist.group.WORLD)
            assert library_comm_world() == world
            model = NativeModel(p, library_comm=True)
        else:
            model = NativeModel(p, dist_group=dist.group.WORLD)
            assert model._dp_active()
        names = [t[0] for t in model.tensors]
        model.training(False)
        loss = model.forwardBackward(mine)
        used_async_bucket = False
        model.update()
        model.synchronize()
        g_dp, w_dp = flat(model.get_gradients_dict(), names), flat(model.get_parameters_dict(), names)
    else:
        model = Model(p, dist_group=dist.group.WORLD)
        assert model._dp_active(), "the data-parallel branch must be live (world > 1 or VD_FORCE_ALLREDUCE=1)"
        model.wrapper.evaluate()                                             # no dropout noise: deterministic comparison
        model.wrapper.zeroGradParameters()
        loss = model.forwardBackward(mine)
        used_async_bucket = model._enc_bucket_work is not None
        model.update()
        torch.cuda.synchronize()
        g_dp = model.wrapperdW.cpu().numpy().copy()     # after update(): summed over ranks, x 1/world, clamped (clamp_adam writes it back)
        w_dp = model.wrapperW.cpu().numpy().copy()
    losses = [None] * world
    dist.all_gather_object(losses, float(loss))
    if rank == 0:
        pb = derive(small_params(batchSize=per_rank * world, gpuid=dev, rank=0))
        if native:
            big = NativeModel(pb)
            big.training(False)
            loss_big = big.forwardBackward(full)
            g_big = flat(big.get_gradients_dict(), names)
            big.update()
            big.synchronize()
            w_big = flat(big.get_parameters_dict(), names)
        else:
            big = Model(pb)
            big.wrapper.evaluate()
            big.wrapper.zeroGradParameters()
            loss_big = big.forwardBackward(full)
            g_big = big.wrapperdW.cpu().numpy().copy()
            big.update()
            torch.cuda.synchronize()
            w_big = big.wrapperW.cpu().numpy()
        assert abs(np.mean(losses) - loss_big) < 1e-5, (losses, loss_big)
        err = np.linalg.norm(g_dp - np.clip(g_big, -5, 5)) / np.linalg.norm(g_big)
        assert err < 1e-5, err
        # Adam's first step is ~lr*sign(g): entries with |g| below the fp32 summation noise may flip sign
        settled = np.abs(g_big) > 1e-6
        assert np.abs(w_dp - w_big)[settled].max() < 1e-6
        assert np.mean(np.abs(w_dp - w_big) < 1e-6) > 0.999
        stats = ''
        if host == 'native-lib':
            from visdial_amd.parallel import library_comm_stats
            stats = ' comm=%s' % (library_comm_stats(),)
        print("DP_GPU_OK world=%d backend=%s host=%s devices=%s async_encoder_bucket=%s grad_rel_err=%.2e%s" % (
            world, backend, host, 'distinct' if os.environ.get('VD_TEST_DISTINCT_DEVICES') == '1' else 'shared', used_async_bucket, err, stats))
    if host == 'native-lib':
        from visdial_amd.parallel import destroy_library_comm
        model.synchronize()
        destroy_library_comm()
    dist.barrier()
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
