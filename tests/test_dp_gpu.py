"""The HIP `Model`'s data-parallel branch on real hardware: bucketed gradient all-reduce through a
process group, checked against the single-process step on the concatenated batch.
  * world 1 over RCCL (backend nccl) with VD_FORCE_ALLREDUCE=1 -- the collectives really run;
  * world 2, both ranks on the one GPU of the test box: RCCL refuses duplicate devices, so the group is gloo and
    the gradient is staged through host memory (parallel.reduce_gradients) -- exercises sharding + both buckets."""
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(nproc, backend, extra_env):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, VD_TEST_BACKEND=backend, HSA_ENABLE_IPC_MODE_LEGACY='0', **extra_env)
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(nproc),
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.join(ROOT, 'tests', 'dp_gpu_worker.py')]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-6000:] + out.stderr[-12000:]
    assert 'DP_GPU_OK' in out.stdout, out.stdout[-2000:]
    return out.stdout


def test_world1_rccl_forced_allreduce():
    out = _run(1, 'nccl', dict(VD_FORCE_ALLREDUCE='1', NCCL_DEBUG='VERSION'))
    assert 'async_encoder_bucket=True' in out


def test_world2_shared_gpu_gloo():
    out = _run(2, 'gloo', {})
    assert 'world=2' in out


def test_native_host_world1_rccl_forced_allreduce():
    """the model-level ABI host (visdial_amd.native): the library hands out wrapperdW + its stream, the collective is
    torch.distributed's (RCCL), ordered behind the step through an external-stream wrapper"""
    out = _run(1, 'nccl', dict(VD_FORCE_ALLREDUCE='1', VD_TEST_HOST='native'))
    assert 'host=native' in out


def test_native_host_world2_shared_gpu_gloo():
    out = _run(2, 'gloo', dict(VD_TEST_HOST='native'))
    assert 'world=2' in out and 'host=native' in out


def test_native_host_library_rccl_world1():
    """the collective BEHIND the C ABI (include/visdial_hip.h: vd_comm_unique_id / vd_comm_init /
    vd_model_allreduce_grads, csrc/comm.hip): the library dlopens librccl, owns the communicator and the communication
    stream, reduces the encoder bucket behind ev_enc_grads and the tail behind the main stream; the host (here Python,
    in production lua/model.lua:initComm) only carries the 128-byte token.  World 1 with the all-reduce forced: the
    RCCL kernels really run on the gradient buffer and the step must equal the plain single-process step."""
    out = _run(1, 'gloo', dict(VD_FORCE_ALLREDUCE='1', VD_TEST_HOST='native-lib', NCCL_DEBUG='VERSION'))
    assert 'host=native-lib' in out


def test_native_host_library_rccl_world2_two_devices(capsys):
    """the FIRST time a box with two GPUs runs this suite, `vd_model_allreduce_grads` meets a real peer -- rank r on
    cuda:r, the library's own RCCL communicator over xGMI, both buckets -- and the two-rank step must equal the big-batch step (gradient
    rel-L2 1e-5, post-Adam parameters).  One-GPU boxes skip, with the reason printed."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    n = torch.cuda.device_count()
    if n < 2:
        with capsys.disabled():
            print("\n[test_dp_gpu] %d GPU visible: the two-device library-RCCL step is NOT exercised on this box "
                  "(world-2 coverage here = gloo + host staging only)" % n)
        pytest.skip("needs two GPUs (found %d): vd_model_allreduce_grads has no real peer on this box" % n)
    out = _run(2, 'gloo', dict(VD_TEST_HOST='native-lib', VD_TEST_DISTINCT_DEVICES='1', NCCL_DEBUG='VERSION'))
    assert 'world=2' in out and 'host=native-lib' in out and 'devices=distinct' in out
    assert "'overlapped': True" in out                     # the encoder bucket went out early


@pytest.mark.parametrize("world", [1, 2])
def test_bench_line_under_torchrun(world):
    """bench.py exactly as the driver launches it for N > 1 (`python -m torch.distributed.run ... bench.py --gpus N --steps K --warmup W`): one JSON
    line from rank 0, exit code 0 on every rank, no probe jobs.  World 1 runs the library's RCCL communicator; world 2 on a one-GPU box is the
    VD_BENCH_SHARE_GPU dry run (every rank on cuda:0, gloo, gradients through host memory: the sharding, the barriers, the max-over-ranks clock
    and the line are exercised, the number means nothing)."""
    import json
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
    if world > 1 and torch.cuda.device_count() < world:
        env['VD_BENCH_SHARE_GPU'] = '1'
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(world), '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.join(ROOT, 'bench.py'), '--gpus', str(world), '--steps', '4', '--warmup', '2',
           '--no-alt', '--no-other-configs', '--no-cpu-baseline']
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-8000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d['n_gpus'] == world and d['steps'] == 4 and d['scaling'] == 'weak' and d['unit'] == 'QA-rounds/s' and d['higher_is_better'] is True
    assert d['config']['global_batch_dialogs'] == 20 * world and d['config']['parallelism'] == 'dp%d' % world
    assert abs(d['value'] - world * 200 * 4 / (d['ms_per_step'] * 4e-3)) < 0.01 * d['value']          # whole-job rate = all ranks' rounds / max-over-ranks time
    assert d['roofline']['frac'] > 0.05 and d['cpu_baseline'] is None
    assert ('hardware queues: HIP default' in out.stderr) == (world > 1)
