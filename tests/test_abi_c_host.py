"""A host with NO tensor library drives the model-level C ABI: examples/host_c_train.c makes the call sequence of
lua/model.lua (create, init, upload, software-pipelined trainIteration, commUniqueId / initComm, retrieveBatch) from plain C
-- dlopen + include/visdial_hip.h + host pointers, i.e. what LuaJIT's ffi does.  No Lua interpreter exists here, so this is the
executable stand-in for the Lua host.  CPU: the header is valid C and the example builds against it.  GPU: its losses, learning
rate and ranks equal the Python host's on the same batches (same library, same calls: bit for bit)."""
import os
import struct
import subprocess

import numpy as np
import pytest

from conftest import ROOT, small_params
from visdial_amd.opts import derive

SRC = os.path.join(ROOT, 'examples', 'host_c_train.c')
SRC_PLUGIN = os.path.join(ROOT, 'examples', 'host_c_plugin_lf_ques.c')
SRC_PLUGIN_MN = os.path.join(ROOT, 'examples', 'host_c_plugin_mn_att.c')
SRC_PLUGIN_GEN = os.path.join(ROOT, 'examples', 'host_c_plugin_lf_ques_gen.c')
SRC_PLUGIN_HRE = os.path.join(ROOT, 'examples', 'host_c_plugin_hre.c')
SRC_PLUGIN_GRAPH = os.path.join(ROOT, 'examples', 'host_c_plugin_graph.c')


def build(tmp_path, src=SRC):
    exe = str(tmp_path / os.path.basename(src)[:-2])
    r = subprocess.run(['gcc', '-O2', '-Wall', '-Werror', '-std=c99', '-D_DEFAULT_SOURCE', '-I', os.path.join(ROOT, 'include'), '-I', os.path.join(ROOT, 'examples'), src,
                        '-ldl', '-lm',
                        '-o', exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return exe


def test_header_is_valid_c_and_the_c_host_builds(tmp_path):
    build(tmp_path, SRC_PLUGIN)
    build(tmp_path, SRC_PLUGIN_MN)
    build(tmp_path, SRC_PLUGIN_GEN)
    build(tmp_path, SRC_PLUGIN_HRE)
    build(tmp_path, SRC_PLUGIN_GRAPH)
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


@pytest.mark.gpu
def test_c_plugin_pair_on_the_operator_level_abi_equals_the_library(tmp_path):
    """examples/host_c_plugin_lf_ques.c composes encoders/lf-ques.lua + decoders/disc.lua + criterion + clamp/adam from
    OPERATOR-LEVEL entry points (module objects with forward / backward, flat parameter vectors) -- what a Lua plug-in file would
    do through ffi.  Its loss, every gradient tensor and the post-Adam parameters must equal the library's own model-level
    implementation of the same pair (NativeModel) on the same parameters and batch."""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from visdial_amd import _lib
    from visdial_amd.dataloader import SyntheticDataloader
    from visdial_amd.native import NativeModel
    p = derive(small_params(encoder='lf-ques', decoder='disc'))
    batch = SyntheticDataloader(p, seed=8).getTrainBatch(p)
    m = NativeModel(dict(p), init_seed=3)
    m.training(False)
    P = m.get_parameters_dict()
    names = [t[0] for t in m.tensors]
    assert names == ['embed', 'ques1.W', 'ques1.b', 'ques2.W', 'ques2.b', 'fuse.W', 'fuse.b', 'opt.W', 'opt.b']
    B, R, Tq = batch['ques_fwd'].shape
    O, To = batch['options'].shape[1], batch['options'].shape[2]
    inp, outp = str(tmp_path / 'in.bin'), str(tmp_path / 'out.bin')
    with open(inp, 'wb') as f:
        f.write(struct.pack('<8i', p['vocabSize'], p['embedSize'], p['rnnHiddenSize'], B, R, O, Tq, To))
        for k in names:
            f.write(np.ascontiguousarray(P[k], np.float32).tobytes())
        for k in ('ques_fwd', 'options', 'answer_ind'):
            f.write(np.ascontiguousarray(batch[k], np.int32).tobytes())
    exe = build(tmp_path, SRC_PLUGIN)
    r = subprocess.run([exe, _lib.LIB_PATH, inp, outp], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    raw = np.fromfile(outp, np.float32)
    n = sum(P[k].size for k in names)
    assert raw.size == 1 + 2 * n + B * R
    ranks = m.retrieveBatch(batch, useGt=True)                         # initial parameters, evaluate mode: same kernels -> equal ranks
    np.testing.assert_array_equal(raw[1 + 2 * n:].astype(np.int64), np.asarray(ranks).reshape(-1))
    loss = m.forwardBackward(batch)
    G = m.get_gradients_dict()
    m.update()
    W1 = m.get_parameters_dict()
    assert abs(float(raw[0]) - loss) < 1e-6 * max(1.0, abs(loss))
    o = 1
    for k in names:
        g = raw[o:o + P[k].size].reshape(P[k].shape)
        ref = G[k]
        den = max(float(np.linalg.norm(ref)), 1e-12)
        assert float(np.linalg.norm(g - ref)) / den < 1e-5, k          # same kernels; float-atomic sums differ in the last bits
        o += P[k].size
    for k in names:
        w = raw[o:o + P[k].size].reshape(P[k].shape)
        settled = np.abs(G[k]) > 1e-6                                  # Adam's first step is ~lr * sign(g)
        assert np.abs(w - W1[k])[settled].max() < 1e-6 if settled.any() else True, k
        o += P[k].size
    m.close()


@pytest.mark.gpu
def test_c_plugin_flagship_pair_on_the_operator_level_abi_equals_the_library(tmp_path):
    """examples/host_c_plugin_mn_att.c composes encoders/mn-att-ques-im-hist.lua (text branches, memory attention, stacked image
    attention, output layer) + decoders/disc.lua from OPERATOR-LEVEL entry points, module object by module object -- the flagship
    pair as a plug-in author would write it (lua/encoders/mn-att-ques-im-hist.lua is the same file in Lua).  Loss, every gradient
    tensor and the post-Adam parameters must equal the library's own model-level implementation on the same parameters and batch;
    the training-mode run (seven Dropout nodes drawing masks) must produce a finite, different loss and finite gradients."""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from visdial_amd import _lib
    from visdial_amd.dataloader import SyntheticDataloader
    from visdial_amd.native import NativeModel
    p = derive(small_params(encoder='mn-att-ques-im-hist', decoder='disc'))
    batch = SyntheticDataloader(p, seed=8).getTrainBatch(p)
    m = NativeModel(dict(p), init_seed=3)
    m.training(False)
    P = m.get_parameters_dict()
    names = [t[0] for t in m.tensors]
    assert names == ['embed', 'hist1.W', 'hist1.b', 'hist2.W', 'hist2.b', 'ques1.W', 'ques1.b', 'ques2.W', 'ques2.b', 'mn1.W', 'mn1.b',
                     'mn2.W', 'mn2.b', 'img_proj.W', 'img_proj.b', 'img_common.W', 'img_common.b', 'ques_common.W', 'ques_common.b',
                     'att.W', 'att.b', 'out.W', 'out.b', 'opt.W', 'opt.b']
    B, R, Tq = batch['ques_fwd'].shape
    Th, O, To = batch['hist'].shape[2], batch['options'].shape[1], batch['options'].shape[2]
    inp, outp = str(tmp_path / 'in.bin'), str(tmp_path / 'out.bin')
    with open(inp, 'wb') as f:
        f.write(struct.pack('<12i', p['vocabSize'], p['embedSize'], p['rnnHiddenSize'], p['imgFeatureSize'], p['imgSpatialSize'],
                            p['commonEmbeddingSize'], B, R, O, Tq, Th, To))
        for k in names:
            f.write(np.ascontiguousarray(P[k], np.float32).tobytes())
        for k, dt in (('ques_fwd', np.int32), ('hist', np.int32), ('img_feat', np.float32), ('options', np.int32), ('answer_ind', np.int32)):
            f.write(np.ascontiguousarray(batch[k], dt).tobytes())
    exe = build(tmp_path, SRC_PLUGIN_MN)
    r = subprocess.run([exe, _lib.LIB_PATH, inp, outp], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    raw = np.fromfile(outp, np.float32)
    n = sum(P[k].size for k in names)
    assert raw.size == 1 + 2 * n
    loss = m.forwardBackward(batch)
    G = m.get_gradients_dict()
    m.update()
    W1 = m.get_parameters_dict()
    assert abs(float(raw[0]) - loss) < 1e-6 * max(1.0, abs(loss))
    o = 1
    for k in names:
        g = raw[o:o + P[k].size].reshape(P[k].shape)
        ref = G[k]
        den = float(np.linalg.norm(ref))
        if k == 'att.b':                                               # the softmax is shift-invariant: the true gradient is 0,
            assert float(np.abs(g).max()) < 1e-6 and float(np.abs(ref).max()) < 1e-6, k      # both sides hold fp32 rounding noise
        else:
            assert float(np.linalg.norm(g - ref)) / den < 1e-5, k      # same kernels; float-atomic sums differ in the last bits
        o += P[k].size
    for k in names:
        w = raw[o:o + P[k].size].reshape(P[k].shape)
        settled = np.abs(G[k]) > 1e-6                                  # Adam's first step is ~lr * sign(g)
        assert np.abs(w - W1[k])[settled].max() < 1e-6 if settled.any() else True, k
        o += P[k].size
    m.close()
    # training mode: the Dropout masks go through the fused gather / attention kernels
    out2 = str(tmp_path / 'out_train.bin')
    r = subprocess.run([exe, _lib.LIB_PATH, inp, out2, 'train'], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    raw2 = np.fromfile(out2, np.float32)
    assert raw2.size == raw.size and np.isfinite(raw2).all()
    assert abs(float(raw2[0]) - float(raw[0])) > 1e-6 and float(np.abs(raw2[1:1 + n]).max()) > 0


@pytest.mark.gpu
@pytest.mark.parametrize("encoder", ['lf-ques', 'lf-ques-im', 'lf-ques-hist', 'lf-ques-im-hist'])
def test_c_plugin_gen_pairs_on_the_operator_level_abi_equal_the_library(tmp_path, encoder):
    """examples/host_c_plugin_lf_ques_gen.c = BASELINE.json configs[0] (the reference's CPU-runnable `-encoder lf-ques -decoder gen`) and
    configs[1] (`-encoder lf-ques-im-hist -decoder gen`), and the two encoders between them (lf-ques-im, lf-ques-hist), composed from OPERATOR-LEVEL entry points: encoder and decoder LSTM stacks with the
    state hand-off of decoders/gen.lua:30-60 (userPrevOutput / userPrevCell forward, userGradPrevOutput / userGradPrevCell / gradPrevOutput /
    userNextGradCell backward), the late-fusion JoinTable of question / image / history, vocabulary projection + log-softmax + summed NLL.
    Loss (the sum over tokens), every gradient tensor and the post-Adam parameters must equal the library's own model-level implementation
    of the pair on the same parameters and batch."""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from visdial_amd import _lib
    from visdial_amd.dataloader import SyntheticDataloader
    from visdial_amd.native import NativeModel
    p = derive(small_params(encoder=encoder, decoder='gen'))
    batch = SyntheticDataloader(p, seed=8).getTrainBatch(p)
    m = NativeModel(dict(p), init_seed=3)
    m.training(False)
    P = m.get_parameters_dict()
    names = [t[0] for t in m.tensors]
    use_im, use_hist = '-im' in encoder, 'hist' in encoder             # the capability flags of opts.lua:54-59
    hist_names = ['hist1.W', 'hist1.b', 'hist2.W', 'hist2.b'] if use_hist else []
    assert names == ['embed', 'ques1.W', 'ques1.b', 'ques2.W', 'ques2.b'] + hist_names + ['fuse.W', 'fuse.b', 'dec1.W', 'dec1.b', 'dec2.W',
                                                                                       'dec2.b', 'vocab.W', 'vocab.b']
    B, R, Tq = batch['ques_fwd'].shape
    Ta = batch['answer_in'].shape[2]
    F = p['imgFeatureSize'] if use_im else 0
    Th = batch['hist'].shape[2] if use_hist else 0
    inp, outp = str(tmp_path / 'in.bin'), str(tmp_path / 'out.bin')
    with open(inp, 'wb') as f:
        f.write(struct.pack('<10i', p['vocabSize'], p['embedSize'], p['rnnHiddenSize'], B, R, Tq, Ta, int(use_im) + 2 * int(use_hist), F, Th))
        for k in names:
            f.write(np.ascontiguousarray(P[k], np.float32).tobytes())
        f.write(np.ascontiguousarray(batch['ques_fwd'], np.int32).tobytes())
        if use_im:
            assert batch['img_feat'].shape == (B, F)
            f.write(np.ascontiguousarray(batch['img_feat'], np.float32).tobytes())
        if use_hist:
            f.write(np.ascontiguousarray(batch['hist'], np.int32).tobytes())
        for k in ('answer_in', 'answer_out'):
            f.write(np.ascontiguousarray(batch[k], np.int32).tobytes())
    exe = build(tmp_path, SRC_PLUGIN_GEN)
    r = subprocess.run([exe, _lib.LIB_PATH, inp, outp], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    raw = np.fromfile(outp, np.float32)
    n = sum(P[k].size for k in names)
    assert raw.size == 1 + 2 * n
    loss = m.forwardBackward(batch)                                    # the SUM over non-pad tokens (model.lua:33-36)
    G = m.get_gradients_dict()
    m.update()
    W1 = m.get_parameters_dict()
    assert abs(float(raw[0]) - loss) < 1e-5 * max(1.0, abs(loss))
    o = 1
    for k in names:
        g = raw[o:o + P[k].size].reshape(P[k].shape)
        den = max(float(np.linalg.norm(G[k])), 1e-12)
        assert float(np.linalg.norm(g - G[k])) / den < 1e-5, k         # same kernels; float-atomic sums differ in the last bits
        o += P[k].size
    for k in names:
        w = raw[o:o + P[k].size].reshape(P[k].shape)
        settled = np.abs(G[k]) > 1e-6                                  # Adam's first step is ~lr * sign(g)
        assert np.abs(w - W1[k])[settled].max() < 1e-6 if settled.any() else True, k
        o += P[k].size
    m.close()


@pytest.mark.gpu
@pytest.mark.parametrize("encoder", ['hre-ques-im-hist', 'hre-ques-hist', 'hrea-ques-im-hist'])
def test_c_plugin_configs2_pair_hre_on_the_operator_level_abi_equals_the_library(tmp_path, encoder):
    """examples/host_c_plugin_hre.c = BASELINE.json configs[2] (hre-ques-im-hist + disc) composed from OPERATOR-LEVEL entry points:
    history and question LSTM stacks, the image embedding joined to the word embedding through MaskTime, the dialog-level recurrence
    over the rounds with its two row permutations.  Loss, every gradient tensor and the post-Adam parameters must equal the library's
    own model-level implementation of the pair on the same parameters and batch."""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from visdial_amd import _lib
    from visdial_amd.dataloader import SyntheticDataloader
    from visdial_amd.native import NativeModel
    p = derive(small_params(encoder=encoder, decoder='disc'))
    batch = SyntheticDataloader(p, seed=8).getTrainBatch(p)
    m = NativeModel(dict(p), init_seed=3)
    m.training(False)
    P = m.get_parameters_dict()
    names = [t[0] for t in m.tensors]
    use_im, attention = '-im-' in encoder, encoder.startswith('hrea')
    assert names == ['embed', 'hist1.W', 'hist1.b', 'hist2.W', 'hist2.b'] + (['img_embed.W', 'img_embed.b'] if use_im else []) + [
        'ques1.W', 'ques1.b', 'ques2.W', 'ques2.b'] + (['att_q.W', 'att_q.b', 'att_h.W', 'att_h.b'] if attention else []) + [
        'dialog.W', 'dialog.b', 'opt.W', 'opt.b']
    B, R, Tq = batch['ques_fwd'].shape
    Th, O, To = batch['hist'].shape[2], batch['options'].shape[1], batch['options'].shape[2]
    inp, outp = str(tmp_path / 'in.bin'), str(tmp_path / 'out.bin')
    with open(inp, 'wb') as f:
        f.write(struct.pack('<13i', p['vocabSize'], p['embedSize'], p['rnnHiddenSize'], p['imgFeatureSize'], p['imgEmbedSize'], B, R, O, Tq, Th, To,
                            int(use_im), int(attention)))
        for k in names:
            f.write(np.ascontiguousarray(P[k], np.float32).tobytes())
        for k, dt in (('ques_fwd', np.int32), ('img_feat', np.float32), ('hist', np.int32), ('options', np.int32), ('answer_ind', np.int32)):
            if k != 'img_feat' or use_im:
                f.write(np.ascontiguousarray(batch[k], dt).tobytes())
    exe = build(tmp_path, SRC_PLUGIN_HRE)
    r = subprocess.run([exe, _lib.LIB_PATH, inp, outp], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    raw = np.fromfile(outp, np.float32)
    n = sum(P[k].size for k in names)
    assert raw.size == 1 + 2 * n
    loss = m.forwardBackward(batch)
    G = m.get_gradients_dict()
    m.update()
    W1 = m.get_parameters_dict()
    assert abs(float(raw[0]) - loss) < 1e-6 * max(1.0, abs(loss))
    o = 1
    for k in names:
        g = raw[o:o + P[k].size].reshape(P[k].shape)
        if k in ('att_q.W', 'att_q.b', 'att_h.b'):   # hrea: the question score and the history bias shift every score of a row alike -> softmax-invariant, true gradient 0
            assert float(np.abs(g).max()) < 1e-6 and float(np.abs(G[k]).max()) < 1e-6, k
        else:
            den = max(float(np.linalg.norm(G[k])), 1e-12)
            assert float(np.linalg.norm(g - G[k])) / den < 1e-5, k     # same kernels; float-atomic sums differ in the last bits
        o += P[k].size
    for k in names:
        w = raw[o:o + P[k].size].reshape(P[k].shape)
        settled = np.abs(G[k]) > 1e-6                                  # Adam's first step is ~lr * sign(g)
        assert np.abs(w - W1[k])[settled].max() < 1e-6 if settled.any() else True, k
        o += P[k].size
    m.close()


@pytest.mark.gpu
@pytest.mark.parametrize("variant,encoder", [(1, 'mn-ques-hist'), (2, 'mn-ques-im-hist'), (3, 'lf-att-ques-im-hist')])
def test_c_plugin_graph_siblings_on_the_operator_level_abi_equal_the_library(tmp_path, variant, encoder):
    """examples/host_c_plugin_graph.c composes the three nngraph siblings of the flagship encoder (+ decoders/disc.lua) from
    OPERATOR-LEVEL entry points: text branches, the memory network, JoinTable + Linear + Tanh fusions (qi, qh), the stacked image
    attention.  Loss, every gradient tensor and the post-Adam parameters must equal the library's own model-level implementation."""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from visdial_amd import _lib
    from visdial_amd.dataloader import SyntheticDataloader
    from visdial_amd.native import NativeModel
    p = derive(small_params(encoder=encoder, decoder='disc'))
    batch = SyntheticDataloader(p, seed=8).getTrainBatch(p)
    m = NativeModel(dict(p), init_seed=3)
    m.training(False)
    P = m.get_parameters_dict()
    names = [t[0] for t in m.tensors]
    text = ['embed'] + [n + s for n in ('hist1', 'hist2', 'ques1', 'ques2') for s in ('.W', '.b')]
    mem = ['mn1.W', 'mn1.b', 'mn2.W', 'mn2.b']
    san = [n + s for n in ('img_proj', 'img_common', 'ques_common', 'att', 'out') for s in ('.W', '.b')]
    assert names == text + {1: mem, 2: ['qi.W', 'qi.b'] + mem, 3: ['qh.W', 'qh.b'] + san}[variant] + ['opt.W', 'opt.b']
    B, R, Tq = batch['ques_fwd'].shape
    Th, O, To = batch['hist'].shape[2], batch['options'].shape[1], batch['options'].shape[2]
    inp, outp = str(tmp_path / 'in.bin'), str(tmp_path / 'out.bin')
    with open(inp, 'wb') as f:
        f.write(struct.pack('<14i', variant, p['vocabSize'], p['embedSize'], p['rnnHiddenSize'], p['imgFeatureSize'], p['imgSpatialSize'],
                            p['commonEmbeddingSize'], p['imgFeatureSize'], B, R, O, Tq, Th, To))
        for k in names:
            f.write(np.ascontiguousarray(P[k], np.float32).tobytes())
        for k, dt in (('ques_fwd', np.int32), ('hist', np.int32), ('img_feat', np.float32), ('options', np.int32), ('answer_ind', np.int32)):
            if k != 'img_feat' or variant != 1:
                f.write(np.ascontiguousarray(batch[k], dt).tobytes())
    exe = build(tmp_path, SRC_PLUGIN_GRAPH)
    r = subprocess.run([exe, _lib.LIB_PATH, inp, outp], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    raw = np.fromfile(outp, np.float32)
    n = sum(P[k].size for k in names)
    assert raw.size == 1 + 2 * n
    loss = m.forwardBackward(batch)
    G = m.get_gradients_dict()
    m.update()
    W1 = m.get_parameters_dict()
    assert abs(float(raw[0]) - loss) < 1e-6 * max(1.0, abs(loss))
    o = 1
    for k in names:
        g = raw[o:o + P[k].size].reshape(P[k].shape)
        if k == 'att.b':                                               # the softmax is shift-invariant: the true gradient is 0
            assert float(np.abs(g).max()) < 1e-6 and float(np.abs(G[k]).max()) < 1e-6, k
        else:
            den = max(float(np.linalg.norm(G[k])), 1e-12)
            assert float(np.linalg.norm(g - G[k])) / den < 1e-5, k     # same kernels; float-atomic sums differ in the last bits
        o += P[k].size
    for k in names:
        w = raw[o:o + P[k].size].reshape(P[k].shape)
        settled = np.abs(G[k]) > 1e-6                                  # Adam's first step is ~lr * sign(g)
        assert np.abs(w - W1[k])[settled].max() < 1e-6 if settled.any() else True, k
        o += P[k].size
    m.close()

