"""The Lua host, EXECUTED: lua/model.lua (model-level ABI), lua/model_ops.lua + lua/vdnn.lua + every plug-in file under
lua/encoders and lua/decoders (operator-level ABI) run inside tests/luavm -- a Lua 5.1 evaluator with a LuaJIT-style ffi over
ctypes and a Torch7 tensor stub -- against the real libvisdial_hip.so on the GPU.  Loss, every gradient tensor and the post-Adam
parameters must equal the library's own path driven from Python (1e-6 / 1e-5: same kernels, float-atomic sums), the fp64 oracle
(1e-4) and the committed golden fixtures; trainIteration reproduces the reference's global `runningLoss` and the learning rate."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

from conftest import grad_mismatches, small_params
from lua_host import LuaHost, first
from luavm import to_lua, to_py
from oracle import visdial_oracle as vo
from test_golden import FILES as GOLDEN, load as load_golden
from test_model_gpu import ALL_ENC, CASES, fuse_masks, make_masks, rel
from visdial_amd.dataloader import SyntheticDataloader
from visdial_amd.opts import derive

pytestmark = pytest.mark.gpu
BASELINE_PAIRS = [('lf-ques', 'gen'), ('lf-ques-im-hist', 'gen'), ('hre-ques-im-hist', 'disc'), ('mn-att-ques-im-hist', 'disc')]
ZERO_GRAD = ('att.b', 'att_q.W', 'att_q.b', 'att_h.b')          # softmax shift invariance: the exact gradient is 0


@pytest.fixture(scope="module")
def gpu():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")


def handle(host, m):
    """the vd_model* a Lua Model holds, for reading gradients through the C ABI from the test"""
    return C.c_void_p(host.get(m, 'h').val)


def lib_get(h, tensors, which):
    from visdial_amd import _lib
    out = {}
    for name, shape in tensors:
        a = np.empty(shape, np.float32)
        _lib.call("vd_model_get_tensor", h, name.encode(), which, a.ctypes.data, a.size)
        out[name] = a
    return out


def close_enough(a, b, name, tol=1e-5):
    if name in ZERO_GRAD:
        assert np.abs(a).max() < 1e-6 and np.abs(b).max() < 1e-6, name
        return
    den = max(float(np.linalg.norm(b)), 1e-12)
    assert float(np.linalg.norm(np.asarray(a, np.float64) - b)) / den < tol, name


def masks_for(p, batch, seed):
    if p['encoder'] == 'mn-att-ques-im-hist':
        return make_masks(p, batch, np.random.RandomState(seed))
    return fuse_masks(p, batch, np.random.RandomState(seed))


# ------------------------------------------------------------------------------------------------- lua/model.lua (model level)
@pytest.mark.parametrize("enc,dec", BASELINE_PAIRS + [('hrea-ques-im-hist', 'gen'), ('lf-att-ques-im-hist', 'disc')])
@pytest.mark.parametrize("train_mode", [False, True])
def test_lua_model_step_equals_library_and_oracle(gpu, enc, dec, train_mode):
    from visdial_amd.native import NativeModel
    from visdial_amd import t7
    p = derive(small_params(encoder=enc, decoder=dec))
    batch = SyntheticDataloader(p, seed=11).getTrainBatch(p)
    nat = NativeModel(dict(p), init_seed=5)
    P = nat.get_parameters_dict()
    shapes = [(n, P[n].shape) for n, _, _, _ in nat.tensors]
    masks = masks_for(p, batch, 5) if train_mode else None
    host = LuaHost(p)
    m = host.model()
    # model.wrapperW:copy(savedModel.modelW) (train.lua:79): a flat vector in the REFERENCE's getParameters() order
    flat = t7.named_to_flat(P, nat._entries(), enc)
    w = host.get(m, 'wrapperW')
    assert w.tn == 'Float' and w.a.shape == (flat.size,)
    host.invoke(w, 'copy', host.tensor(flat, 'Float'))
    if masks:
        host.invoke(m, 'setDropoutMasks', to_lua(host.vm, dict(masks)))
        nat.set_dropout_masks(masks)
    else:
        host.invoke(m, 'setMode', False)
        nat.training(False)
    loss = first(host.invoke(m, 'forwardBackward', host.batch(batch)))
    ref_loss = nat.forwardBackward(batch)
    assert abs(loss - ref_loss) <= 1e-6 * max(1.0, abs(ref_loss)), (loss, ref_loss)
    g, g_ref = lib_get(handle(host, m), shapes, 1), nat.get_gradients_dict()
    for k in g_ref:
        close_enough(g[k], g_ref[k], k)
    # the parameters really arrived tensor by tensor (hre-*: img_embed sits behind embed in the reference's order)
    w_lua = lib_get(handle(host, m), shapes, 0)
    for k in P:
        np.testing.assert_array_equal(w_lua[k], P[k], err_msg=k)
    # and against the fp64 restatement
    P64 = {k: v.astype(np.float64) for k, v in P.items()}
    drop = {k: v.astype(np.float64) for k, v in masks.items()} if masks else None
    ref = vo.forward_backward(enc, dec, P64, p, batch, drop)
    assert abs(loss - ref['loss']) < 1e-4 * max(1.0, abs(ref['loss']))
    bad = grad_mismatches(g, ref['grads'])
    assert not bad, bad
    # a read-only access of wrapperW followed by a step leaves the parameters untouched (checkout / commit)
    w2 = to_py(host.get(m, 'wrapperW'))
    np.testing.assert_array_equal(w2, flat)
    nat.close()
    host.close()


@pytest.mark.parametrize("enc,dec", BASELINE_PAIRS)
def test_lua_train_iteration_reproduces_running_loss_and_lr(gpu, enc, dec):
    """model:trainIteration(dataloader) x 3 from Lua (software-pipelined prefetch, GLOBAL runningLoss, optims.learningRate) ==
    NativeModel.trainIteration x 3 on the same sample stream; dropout drawn from the same seed on both sides"""
    from visdial_amd.native import NativeModel
    p = derive(small_params(encoder=enc, decoder=dec))
    nat = NativeModel(dict(p), init_seed=1234)
    host = LuaHost(p)
    m = host.model()
    d_lua = host.dataloader(SyntheticDataloader(p, seed=21))
    d_py = SyntheticDataloader(p, seed=21)
    for it in range(3):
        a = first(host.invoke(m, 'trainIteration', d_lua))
        b = nat.trainIteration(d_py)
        assert abs(a - b) <= 1e-4 * max(1.0, abs(b)), (it, a, b)
        # (the Lua host sets the DOUBLE params.learningRate like the reference's optims table; the library's own default is its float field)
        assert abs(host.get(m, 'optims', 'learningRate') - nat.optims['learningRate']) < 1e-9
    run = host.vm.globals.get('runningLoss')
    assert run > 0 and abs(run - nat.runningLoss) < 1e-5 * max(1.0, nat.runningLoss)
    if dec == 'gen':
        assert run < 0.2 * a                          # the EMA is per token, curLoss is the sum
    shapes = [(n, (c,) if n.endswith('.b') else (r, c)) for n, _, r, c in nat.tensors]
    w_lua, w_py = lib_get(handle(host, m), shapes, 0), nat.get_parameters_dict()
    for k in w_py:
        if k not in ZERO_GRAD:
            assert np.abs(w_lua[k] - w_py[k]).max() < 2e-5, k
    nat.close()
    host.close()


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(f) for f in GOLDEN])
@pytest.mark.parametrize("which", ['model', 'ops'])
def test_lua_hosts_match_the_golden_fixtures(gpu, path, which):
    enc, dec, p, z, P, batch, masks, grads = load_golden(path)
    host = LuaHost(p)
    lua_masks = to_lua(host.vm, dict(masks)) if masks else None
    if which == 'model':
        from visdial_amd import t7
        m = host.model()
        spec = vo.param_spec(enc, dec, p)
        flat = t7.named_to_flat(P, spec, enc)
        host.invoke(host.get(m, 'wrapperW'), 'copy', host.tensor(flat, 'Float'))
        if masks:
            host.invoke(m, 'setDropoutMasks', lua_masks)
        else:
            host.invoke(m, 'setMode', False)
        loss = first(host.invoke(m, 'forwardBackward', host.batch(batch)))
        g = lib_get(handle(host, m), [(n, s) for n, s, _ in spec], 1)
    else:
        m = host.model_ops()
        host.ops_set_parameters(m, P)
        if masks:
            host.vm.call(host.get(m, 'encoder', 'vdnn', 'pinMasks'), lua_masks)
        else:
            host.invoke(m, 'evaluate')
        loss = first(host.invoke(m, 'forwardBackward', host.batch(batch)))
        g = host.ops_get(m, 'dW', {k: v.shape for k, v in P.items()})
    ref = float(z['loss'])
    assert abs(loss - ref) < 1e-4 * max(1.0, abs(ref))
    bad = grad_mismatches(g, grads)
    assert not bad, bad
    if dec == 'disc':
        if which == 'model':
            host.invoke(m, 'setMode', False)
            host.get(m, 'params').set('useGt', False)
        else:
            host.invoke(m, 'evaluate')
            host.get(m, 'params').set('useGt', False)
        ranks = to_py(first(host.invoke(m, 'retrieveBatch', host.batch(batch))))
        assert ranks.dtype == np.float64 and ranks.shape == z['eval.ranks'].shape
        assert (ranks != z['eval.ranks']).mean() < 0.02          # fp32 near ties only (bit-exactness vs device scores: tests below)
    host.close()


# --------------------------------------------------------------------------- lua/model_ops.lua + vdnn.lua + the plug-in files
@pytest.mark.parametrize("enc,dec", [(e, d) for e in ALL_ENC for d in ('disc', 'gen')])
def test_lua_composed_plugins_equal_library_and_oracle(gpu, enc, dec):
    """every encoder file x decoder file, composed IN LUA from vdnn module objects: evaluate mode against the library's model-level
    implementation and the oracle; then training mode with the Dropout noise pinned by site name, against the oracle"""
    from visdial_amd.native import NativeModel
    p = derive(small_params(encoder=enc, decoder=dec))
    batch = SyntheticDataloader(p, seed=8).getTrainBatch(p)
    nat = NativeModel(dict(p), init_seed=3)
    nat.training(False)
    P = nat.get_parameters_dict()
    shapes = {k: v.shape for k, v in P.items()}
    host = LuaHost(p)
    m = host.model_ops()
    assert [n for n, _ in host.ops_spec(m)] == [t[0] for t in nat.tensors]           # getParameters() order = the library's declaration
    host.ops_set_parameters(m, P)
    host.invoke(m, 'evaluate')
    lua_batch = host.batch(batch)
    before = {k: to_py(v) for k, v in lua_batch.items()}
    out = host.invoke(m, 'forwardBackward', lua_batch)
    for k, v in lua_batch.items():                                                   # the caller's batch is not modified
        np.testing.assert_array_equal(to_py(v), before[k], err_msg=k)
    loss = out[0]
    ref_loss = nat.forwardBackward(batch)
    assert abs(loss - ref_loss) <= 1e-5 * max(1.0, abs(ref_loss)), (loss, ref_loss)
    g, g_ref = host.ops_get(m, 'dW', shapes), nat.get_gradients_dict()
    for k in g_ref:
        close_enough(g[k], g_ref[k], k)
    P64 = {k: v.astype(np.float64) for k, v in P.items()}
    ref = vo.forward_backward(enc, dec, P64, p, batch, None)
    assert abs(loss - ref['loss']) < 1e-4 * max(1.0, abs(ref['loss']))
    bad = grad_mismatches(g, ref['grads'])
    assert not bad, bad
    # ModelOps:trainIteration on the same batch: zeroGradParameters, step, clamp(-5, 5) + adam (optim_updates.lua:62-91)
    fixed = SyntheticDataloader(p, seed=8)
    fixed.getTrainBatch = lambda params, **kw: batch
    host.invoke(m, 'trainIteration', host.dataloader(fixed))
    after = host.ops_get(m, 'W', shapes)
    for k in P:
        w2, _ = vo.clamp_adam(P64[k].reshape(-1), g[k].astype(np.float64).reshape(-1), {}, p['learningRate'])
        settled = np.abs(g[k].reshape(-1)) > 1e-6                                    # Adam's first step is ~lr * sign(g)
        if settled.any():
            assert np.abs(after[k].reshape(-1) - w2)[settled].max() < 1e-6, k
    # training mode, noise pinned per Dropout node
    masks = masks_for(p, batch, 9)
    if masks:
        host.ops_set_parameters(m, P)
        host.invoke(m, 'training')
        host.vm.call(host.get(m, 'encoder', 'vdnn', 'pinMasks'), to_lua(host.vm, dict(masks)))
        host.invoke(host.get(m, 'fp'), 'zeroGrad')
        loss_t = first(host.invoke(m, 'forwardBackward', host.batch(batch)))
        ref_t = vo.forward_backward(enc, dec, P64, p, batch, {k: v.astype(np.float64) for k, v in masks.items()})
        assert abs(loss_t - ref_t['loss']) < 1e-4 * max(1.0, abs(ref_t['loss']))
        assert abs(loss_t - loss) > 1e-7                                              # the masks did something
        bad = grad_mismatches(host.ops_get(m, 'dW', shapes), ref_t['grads'])
        assert not bad, bad
    nat.close()
    host.close()


def test_lua_composed_flagship_at_width(gpu):
    """the flagship pair composed in Lua at H = 512 / 100 options / 14 x 14 regions ('mid'): the production kernels"""
    from visdial_amd.native import NativeModel
    p = derive(small_params(**CASES['mid']))
    batch = SyntheticDataloader(p, seed=4).getTrainBatch(p)
    nat = NativeModel(dict(p), init_seed=3)
    masks = make_masks(p, batch, np.random.RandomState(2))
    nat.set_dropout_masks(masks)
    P = nat.get_parameters_dict()
    host = LuaHost(p)
    m = host.model_ops()
    host.ops_set_parameters(m, P)
    host.vm.call(host.get(m, 'encoder', 'vdnn', 'pinMasks'), to_lua(host.vm, dict(masks)))
    loss = first(host.invoke(m, 'forwardBackward', host.batch(batch)))
    ref_loss = nat.forwardBackward(batch)
    assert abs(loss - ref_loss) <= 1e-5 * max(1.0, abs(ref_loss))
    g, g_ref = host.ops_get(m, 'dW', {k: v.shape for k, v in P.items()}), nat.get_gradients_dict()
    for k in g_ref:
        close_enough(g[k], g_ref[k], k, tol=2e-5)
    # three more steps do not grow device memory: activations are step-scoped (vdnn.releaseStep)
    free0 = torch.cuda.mem_get_info()[0]
    for _ in range(3):
        host.invoke(m, 'forwardBackward', host.batch(batch))
    assert free0 - torch.cuda.mem_get_info()[0] < (64 << 20)
    nat.close()
    host.close()


# ------------------------------------------------------------------------------------------------------------- retrieval paths
@pytest.mark.parametrize("enc,dec", [('mn-att-ques-im-hist', 'disc'), ('lf-ques', 'disc'), ('lf-ques-im-hist', 'gen')])
def test_lua_retrieve_batch_equals_library(gpu, enc, dec):
    from visdial_amd.native import NativeModel
    p = derive(small_params(encoder=enc, decoder=dec))
    dl = SyntheticDataloader(p, seed=13)
    batch = dl.getTrainBatch(p)
    if dec == 'gen':
        dl.add_gen_options(batch, batch['ques_fwd'].shape[0])
    nat = NativeModel(dict(p), init_seed=7)
    nat.training(False)
    P = nat.get_parameters_dict()
    gt_ref, all_ref = nat.retrieveBatch(batch, useGt=True), nat.retrieveBatch(batch, useGt=False)
    host = LuaHost(p)
    m = host.model()
    from visdial_amd import t7
    flat = t7.named_to_flat(P, nat._entries(), enc)
    host.invoke(m, 'setFlatParameters', host.tensor(flat, 'Float'))
    host.invoke(m, 'setMode', False)
    R = p['maxQuesCount']
    for use_gt, ref in ((True, gt_ref), (False, all_ref)):
        host.get(m, 'params').set('useGt', use_gt)
        r = to_py(first(host.invoke(m, 'retrieveBatch', host.batch(batch))))
        assert r.dtype == np.float64                                   # DoubleTensor like the reference (utils.lua:106-128)
        assert r.shape == ((batch['ques_fwd'].shape[0], R) if use_gt else np.asarray(ref).shape)
        np.testing.assert_array_equal(r.reshape(-1), np.asarray(ref).reshape(-1))
    if dec == 'disc':                                                   # and the operator-level host (ModelOps:retrieveBatch)
        mo = host.model_ops()
        host.ops_set_parameters(mo, P)
        host.invoke(mo, 'evaluate')
        for use_gt, ref in ((True, gt_ref), (False, all_ref)):
            host.get(mo, 'params').set('useGt', use_gt)
            r = to_py(first(host.invoke(mo, 'retrieveBatch', host.batch(batch))))
            np.testing.assert_array_equal(r.reshape(-1), np.asarray(ref).reshape(-1))
    nat.close()
    host.close()


def test_lua_split_evaluation_and_comm(gpu):
    """Model:evaluate / :retrieve / :predict over a split, getFlatParameters / setFlatParameters round trip, the rendezvous token
    and vd_comm_init at world 1 -- from Lua"""
    from visdial_amd.native import NativeModel
    p = derive(small_params(encoder='mn-att-ques-im-hist', decoder='disc'))
    nat = NativeModel(dict(p), init_seed=1234)
    host = LuaHost(p)
    m = host.model()
    l_py, _ = nat.evaluate(SyntheticDataloader(p, seed=5, num_threads=5), 'val')
    l_lua = first(host.invoke(m, 'evaluate', host.dataloader(SyntheticDataloader(p, seed=5, num_threads=5)), 'val'))
    assert abs(l_lua - l_py) < 1e-5 * max(1.0, abs(l_py))
    dl = SyntheticDataloader(p, seed=5, num_threads=5)
    D = host.dataloader(dl)
    D.fields['unique_img_val'] = to_lua(host.vm, list(range(101, 106)))
    D.fields['val_num_rounds'] = to_lua(host.vm, [p['maxQuesCount']] * 5)
    recs = to_py(first(host.invoke(m, 'retrieve', D, 'val')))
    assert len(recs) == 5 * p['maxQuesCount'] and recs[0]['image_id'] == 101 and recs[0]['round_id'] == 1
    nat.training(False)
    ranks_py = nat.retrieveBatch(SyntheticDataloader(p, seed=5, num_threads=5).getTestBatch(1, p, 'val')[0], useGt=True)
    assert [r['ranks'] for r in recs[:p['maxQuesCount']]] == [float(x) for x in np.asarray(ranks_py).reshape(-1)[:p['maxQuesCount']]]
    pred = to_py(first(host.invoke(m, 'predict', D, 'val')))
    assert len(pred) == 5 * p['maxQuesCount'] and sorted(pred[0]['ranks']) == [float(i) for i in range(1, p['numOptions'] + 1)]
    flat = to_py(first(host.invoke(m, 'getFlatParameters')))
    host.invoke(m, 'setFlatParameters', host.tensor(flat[::-1].copy(), 'Float'))
    np.testing.assert_array_equal(to_py(first(host.invoke(m, 'getFlatParameters'))), flat[::-1])
    token = first(host.vm.call(host.get(m, 'commUniqueId')))
    assert isinstance(token, str) and len(token) == 128
    host.invoke(m, 'initComm', 0, 1, token)
    assert host.get(m, 'world') == 1
    first(host.invoke(m, 'trainIteration', host.dataloader(SyntheticDataloader(p, seed=3))))
    from visdial_amd import _lib
    _lib.call("vd_comm_destroy")
    nat.close()
    host.close()


@pytest.mark.parametrize("enc", ['lf-ques-im-hist', 'hre-ques-im-hist'])
def test_lua_beam_search_equals_library(gpu, enc):
    """Model:generateAnswers (model.lua:432-613) from Lua: beam search over vd_model_encode / decode_begin / decode_step /
    decode_select gives the sentences of the Python host (which is token-exact against the oracle's beam search)"""
    from test_dataloader_cpu import raw_dataset
    from visdial_amd.dataloader import Dataloader
    from visdial_amd.native import NativeModel
    from visdial_amd.opts import default_params
    rng = np.random.RandomState(2)
    info, raw, img = raw_dataset(rng, n=3, R=3, MQ=5, MA=4, V=20, O=4, nopt=12, F=8)
    raw = {k.replace('_train', '_val'): v for k, v in raw.items()}
    img = {k.replace('_train', '_val'): v for k, v in img.items()}
    info['unique_img_val'] = info.pop('unique_img_train')
    p = derive(default_params(encoder=enc, decoder='gen', embedSize=12, rnnHiddenSize=32, imgFeatureSize=8, imgEmbedSize=8,
                              numLayers=2, batchSize=1, learningRate=1e-3, gpuid=0))
    dl = Dataloader(seed=1).from_arrays(info, raw, img, p, ['val'])
    for k in ('vocabSize', 'maxQuesCount', 'maxQuesLen', 'maxAnsLen'):
        p[k] = getattr(dl, k)
    nat = NativeModel(p, init_seed=1234)
    cfg = dict(beamSize=3, beamLen=6, maxThreads=2)
    ref = nat.generateAnswers(dl, 'val', cfg)
    host = LuaHost(p)
    m = host.model()
    D = host.dataloader(dl)
    D.fields['word2ind'] = to_lua(host.vm, dict(dl.word2ind))
    D.fields['ind2word'] = to_lua(host.vm, {int(k): v for k, v in dl.ind2word.items()})
    D.fields['numThreads'] = to_lua(host.vm, {'val': 3})
    out = to_py(first(host.invoke(m, 'generateAnswers', D, 'val', to_lua(host.vm, cfg))))
    assert len(out) == 2
    for conv in range(2):
        for it in range(len(ref[conv]['dialog'])):
            assert out[conv]['dialog'][it]['answer'].split() == ref[conv]['dialog'][it]['answer'].split(), (conv, it)
    nat.close()
    host.close()
