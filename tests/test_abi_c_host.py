"""A host with NO tensor library drives the model-level C ABI: examples/host_c_train.c makes the call sequence of
lua/model.lua (create, init, upload, software-pipelined trainIteration, commUniqueId / initComm, retrieveBatch) from plain C
-- dlopen + include/visdial_hip.h + host pointers, i.e. what LuaJIT's ffi does.  (The Lua host itself is executed by
tests/test_lua_host_gpu.py inside tests/luavm; the operator-level C twins of the plug-in files that used to live here as its
stand-ins are gone.)  CPU: the header is valid C and the example builds against it.  GPU: its losses, learning
rate and ranks equal the Python host's on the same batches (same library, same calls: bit for bit)."""
import os
import struct
import subprocess

import numpy as np
import pytest

from conftest import ROOT, small_params
from visdial_amd.opts import derive

SRC = os.path.join(ROOT, 'examples', 'host_c_train.c')


def build(tmp_path, src=SRC):
    exe = str(tmp_path / os.path.basename(src)[:-2])
    r = subprocess.run(['gcc', '-O2', '-Wall', '-Werror', '-std=c99', '-D_DEFAULT_SOURCE', '-I', os.path.join(ROOT, 'include'), '-I', os.path.join(ROOT, 'examples'), src,
                        '-ldl', '-lm',
                        '-o', exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return exe


def test_header_is_valid_c_and_the_c_host_builds(tmp_path):
    exe = build(tmp_path)
    # without a library the host fails loudly at dlopen -- no fallback of any kind
    r = subprocess.run([exe, '/nonexistent/libvisdial_hip.so', '/dev/null', '1', '0'], capture_output=True, text=True)
    assert r.returncode == 2 and 'dlopen' in r.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("use_comm", [0, 1])
def test_c_host_equals_python_host(tmp_path, use_comm):
    import torch
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from visdial_amd import _lib
    from visdial_amd.dataloader import SyntheticDataloader
    from visdial_amd.native import NativeModel
    p = derive(small_params())
    dl = SyntheticDataloader(p, seed=31)
    batches = [dl.getTrainBatch(p) for _ in range(3)]
    B, R, Tq = batches[0]['ques_fwd'].shape
    Th, To, O = batches[0]['hist'].shape[2], batches[0]['options'].shape[2], batches[0]['options'].shape[1]
    # (every batch of the file must have the same trimmed lengths: the C host reads fixed-size records)
    batches = [b for b in batches if b['ques_fwd'].shape[2] == Tq and b['hist'].shape[2] == Th and b['options'].shape[2] == To]
    assert batches
    img_floats = int(np.prod(batches[0]['img_feat'].shape[1:]))
    path = str(tmp_path / 'batches.bin')
    with open(path, 'wb') as f:
        f.write(struct.pack('<14i', B, R, O, Tq, Th, To, img_floats, p['vocabSize'], p['embedSize'], p['rnnHiddenSize'],
                            p['imgFeatureSize'], p['imgSpatialSize'], p['commonEmbeddingSize'], len(batches)))
        for b in batches:
            for k, dt in (('ques_fwd', np.int32), ('hist', np.int32), ('img_feat', np.float32), ('options', np.int32),
                          ('answer_ind', np.int32)):
                f.write(np.ascontiguousarray(b[k], dtype=dt).tobytes())
    steps = 5
    exe = build(tmp_path)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
    r = subprocess.run([exe, _lib.LIB_PATH, path, str(steps), str(use_comm)], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    out = r.stdout.split('\n')
    c_loss = [float(l.split()[2]) for l in out if l.startswith('loss ')]
    c_lr = [float(l.split()[1]) for l in out if l.startswith('lr ')][0]
    c_rank = [int(l.split()[2]) for l in out if l.startswith('rank ')]
    assert len(c_loss) == steps and len(c_rank) == B * R
    # the same calls from Python (visdial_amd.native drives the same entry points through ctypes)
    m = NativeModel(dict(p, learningRate=1e-3, lrDecayRate=0.9997592083, minLRate=5e-5, dropout=0.5), init_seed=1234)
    m.training(False)
    ranks = m.retrieveBatch(batches[0], useGt=True)           # initial parameters: no atomics upstream -> exact
    np.testing.assert_array_equal(np.asarray(c_rank), np.asarray(ranks).reshape(-1))
    m.upload(batches[0])
    py_loss = []
    for it in range(steps):
        _lib.call("vd_model_forward_backward", m.h, 0)
        _lib.call("vd_model_update", m.h, 1.0)
        m.upload(batches[(it + 1) % len(batches)])
        py_loss.append(m.loss())
    assert np.float32(c_loss[0]) == np.float32(py_loss[0])     # the first loss precedes every float-atomic sum: bit for bit
    # later steps see parameters that went through float-atomic weight-gradient sums (order-dependent in the last bits)
    assert np.allclose(c_loss, py_loss, rtol=1e-4, atol=0), (c_loss, py_loss)
    assert abs(c_lr - m.optims['learningRate']) < 1e-12
    assert c_loss[-1] < c_loss[0]          # and it trains
    m.close()
