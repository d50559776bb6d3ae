"""Fixtures made by EXECUTING the reference's unedited train.lua END TO END under tests/luavm (tests/golden/make_reference_train_golden.py):
opts.lua -> dataloader.lua on a small raw dataset -> Model -> four training iterations with dropout on -> torch.save of the checkpoints,
then the unedited evaluate.lua on the epoch-2 checkpoint and the val split (-useGt true: Model:retrieve, false: Model:predict, -saveRanks)
and, for the gen pair, the unedited generate.lua (beam search over two val dialogs -> results.json).
Stored: the raw dataset, the initial flat vector (wrapper:getParameters() order), the thread ids and the Dropout noise of every iteration,
and where the run ended (flat vector after epoch 1 and at the end, learning rate, runningLoss).

CPU: the oracle, fed by the PRODUCT loader and splitting the flat vector with t7.reference_order, replays the run to 1e-9 (the fixture
stores fp32 deltas).  GPU: both HIP hosts run the same four iterations.  Nothing here reads /root/reference."""
import glob
import json
import os

import numpy as np
import pytest
import torch

from conftest import ROOT
from oracle import visdial_oracle as vo
from visdial_amd import t7
from visdial_amd.dataloader import Dataloader
from visdial_amd.opts import derive

FILES = sorted(glob.glob(os.path.join(ROOT, 'tests', 'golden', 'ref_train__*.npz')))
IDS = [os.path.basename(f)[len('ref_train__'):-4] for f in FILES]


def load(path):
    z = np.load(path)
    get = lambda pre: {k[len(pre):]: z[k] for k in z.files if k.startswith(pre)}
    opt = derive(json.loads(str(z['opt.json'])))
    dl = Dataloader(seed=1).from_arrays(json.loads(str(z['info.json'])), get('raw.'), get('img.'), opt, ['train', 'val'])
    spec = vo.param_spec(opt['encoder'], opt['decoder'], opt)

    def batch(it):
        ids = z['ids'][it]
        b = dl.getIndexData(ids, opt, 'train')
        if opt['decoder'] == 'disc':                                  # dataloader.lua:330-337
            o = dl.getIndexOption(ids, opt, 'train')
            b['options'] = o.reshape(o.shape[0] * o.shape[1], o.shape[2], -1)
            b['answer_ind'] = b['answer_ind'].reshape(-1)
        return b
    masks = lambda it: get('it%d.mask.' % it)
    batch.loader = dl
    return z, opt, spec, batch, masks


def _gen_batches(z, opt):
    get = lambda pre: {k[len(pre):]: z[k] for k in z.files if k.startswith(pre)}
    gopt = dict(opt, concatHistory=False, maxHistoryLen=60)
    dl = Dataloader(seed=1).from_arrays(json.loads(str(z['info.json'])), get('raw.'), get('img.'), gopt, ['val'])
    return [dl.getIndexData(np.array([c]), gopt, 'val') for c in (1, 2)]


def split(flat, spec, enc):
    out, o = {}, 0
    for n, shape, _ in t7.reference_order(enc, spec):
        k = int(np.prod(shape))
        out[n] = np.asarray(flat[o:o + k], np.float64).reshape(shape)
        o += k
    assert o == len(flat)
    return out


def test_fixture_set():
    assert sorted(IDS) == ['lf-ques-im-hist__gen', 'mn-att-ques-im-hist__disc']


@pytest.mark.parametrize("path", FILES, ids=IDS)
def test_oracle_replays_the_executed_train_lua(path):
    z, opt, spec, batch, masks = load(path)
    enc, dec = opt['encoder'], opt['decoder']
    P = split(z['W0'], spec, enc)
    state, lr, rl = {}, opt['learningRate'], 0.0
    for it in range(4):
        b = batch(it)
        m = {k: v.astype(np.float64) for k, v in masks(it).items()}
        P, r = vo.train_iteration(enc, dec, P, opt, b, m or None, state, lr)
        if lr > opt['minLRate']:
            lr *= opt['lrDecayRate']
        cur = r['loss'] / max(int((b['answer_out'] > 0).sum()), 1) if dec == 'gen' else r['loss']
        rl = 0.95 * rl + 0.05 * cur if rl > 0 else cur
        if it == 1:
            flat = np.concatenate([P[n].reshape(-1) for n, _, _ in t7.reference_order(enc, spec)])
            assert (np.abs((flat - z['W0']) - z['epoch1.delta']) < 1e-9 + 1e-7 * np.abs(z['epoch1.delta'])).all()      # model_epoch_1.t7 (fp32 deltas)
    flat = np.concatenate([P[n].reshape(-1) for n, _, _ in t7.reference_order(enc, spec)])
    assert (np.abs((flat - z['W0']) - z['final.delta']) < 1e-9 + 1e-7 * np.abs(z['final.delta'])).all()     # model_epoch_2.t7 == the final state
    assert abs(lr - float(z['end'][0])) < 1e-15 and abs(rl - float(z['end'][1])) < 1e-9 * max(1.0, abs(rl))
    assert np.abs(z['final.delta']).max() > 1e-3                                     # (four Adam steps did move the weights)
    # evaluate.lua on that checkpoint: val split in batches of 2 (dataloader.lua:342-375), retrieveBatch + utils.computeRanks
    val, start, gt_r, all_r = batch.loader, 1, [], []
    while start <= 5:
        b, start = val.getTestBatch(start, dict(opt, batchSize=2), 'val')
        sc = vo.retrieve(enc, dec, P, opt, b)
        gt_r.append(vo.compute_ranks(sc, np.asarray(b['answer_ind']).reshape(-1) - 1).reshape(-1, 10))
        all_r.append(vo.compute_ranks(sc).reshape(-1, 10, sc.shape[1]))
    np.testing.assert_array_equal(np.concatenate(gt_r), z['eval.gt_ranks'])
    np.testing.assert_array_equal(np.concatenate(all_r), z['eval.all_ranks'])
    if dec == 'gen':          # generate.lua: per-round history (it never sets concatHistory), beams of the first two val dialogs
        bs, bl, START, END = (int(v) for v in z['gen.params'])
        Pg = dict(P)
        Pg['vocab.b'] = z['gen.vocab_b'].astype(np.float64)
        for c, gb in enumerate(_gen_batches(z, opt)):
            got = vo.generate_beam(enc, Pg, dict(opt, concatHistory=False, maxHistoryLen=60), gb, bs, bl, START, END)
            np.testing.assert_array_equal(np.array([t for t, _ in got]), z['gen.tokens'][c])


@pytest.mark.gpu
@pytest.mark.parametrize("path", FILES, ids=IDS)
def test_hip_hosts_replay_the_executed_train_lua(path):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from visdial_amd.model import Model
    from visdial_amd.native import NativeModel
    z, opt, spec, batch, masks = load(path)
    enc = opt['encoder']
    W0 = {k: v.astype(np.float32) for k, v in split(z['W0'], spec, enc).items()}
    want = split(z['final.delta'], spec, enc)

    class Loader(object):
        def __init__(self):
            self.i = 0

        def getTrainBatch(self, params, **_k):
            b = batch(min(self.i, 3))
            self.i += 1
            return b
    for host in (Model(dict(opt)), NativeModel(dict(opt))):
        host.set_parameters_dict(W0)
        (host.wrapper.training if hasattr(host, 'wrapper') else (lambda: host.training(True)))()
        dl = Loader()
        host.runningLoss = 0
        for it in range(4):
            host.set_dropout_masks(masks(it))                 # the noise the reference's Dropout nodes drew in that iteration
            host.trainIteration(dl)
        W = host.get_parameters_dict()
        for k in W0:
            got = W[k].astype(np.float64) - W0[k].astype(np.float64)
            err = np.abs(got - want[k])
            assert err.max(initial=0.0) <= 4.0001 * opt['learningRate'] + 4e-7 * np.abs(W0[k]).max(), k
            if np.abs(want[k]).max() < 1e-9:
                continue
            assert np.linalg.norm(err) <= 0.02 * np.linalg.norm(want[k]) + 1e-6, (k, float(np.linalg.norm(err)), float(np.linalg.norm(want[k])))
        assert abs(host.optims['learningRate'] - float(z['end'][0])) < 1e-7 * opt['learningRate']
        assert abs(host.runningLoss - float(z['end'][1])) < 1e-4 * max(1.0, abs(float(z['end'][1])))
        # evaluate.lua's two modes on the REFERENCE's final weights, through the product's split loops (visdial_amd/split_eval.py)
        host.set_parameters_dict({k: (W0[k].astype(np.float64) + want[k]).astype(np.float32) for k in W0})
        host.set_dropout_masks(None)
        host.params['batchSize'] = 2
        # (the 100 candidates of a round are drawn from a 40-entry option list: exact duplicates tie, and fp32 may order a tie group
        #  differently -- a rank may move by at most the number of candidates whose fp64 score lies within 1e-3 of the candidate's own)
        Pf = {k: W0[k].astype(np.float64) + want[k] for k in W0}
        start, sc = 1, []
        while start <= 5:
            b, start = batch.loader.getTestBatch(start, dict(opt, batchSize=2), 'val')
            sc.append(vo.retrieve(enc, opt['decoder'], Pf, opt, b).reshape(-1, 10, 100))
        sc = np.concatenate(sc)
        near = (np.abs(sc[..., :, None] - sc[..., None, :]) < 1e-3 * np.maximum(1.0, np.abs(sc[..., :, None]))).sum(-1) - 1      # [5, 10, 100]
        metrics, recs = host.retrieve(batch.loader, 'val')
        got = np.array([r['ranks'] for r in recs]).reshape(5, 10)
        gt_opt = np.argmax(z['eval.all_ranks'] == z['eval.gt_ranks'][..., None], axis=-1)                   # which candidate is the ground truth
        slack = np.take_along_axis(near, gt_opt[..., None], -1)[..., 0]
        assert (np.abs(got - z['eval.gt_ranks']) <= slack).all()
        assert [(r['image_id'], r['round_id']) for r in recs] == [(100 + i, j + 1) for i in range(5) for j in range(10)]
        pred = host.predict(batch.loader, 'val')
        got = np.array([r['ranks'] for r in pred]).reshape(5, 10, -1)
        assert (np.abs(got - z['eval.all_ranks']) <= near).all()
        assert (got == z['eval.all_ranks']).mean() > 0.5
        host.params['batchSize'] = opt['batchSize']
        if opt['decoder'] == 'gen':          # generate.lua's beams, through the product's host loop
            bs, bl, START, END = (int(v) for v in z['gen.params'])
            Pg = {k: (W0[k].astype(np.float64) + want[k]).astype(np.float32) for k in W0}
            Pg['vocab.b'] = z['gen.vocab_b'].astype(np.float32)
            host.set_parameters_dict(Pg)
            gbs = _gen_batches(z, opt)
            words = {int(i): w for w, i in json.loads(str(z['info.json']))['word2ind'].items()}
            words[START], words[END] = '<START>', '<END>'

            class GenLoader(object):
                word2ind, ind2word, numThreads, unique_img_val = {'<START>': START, '<END>': END}, words, {'val': 2}, [100, 101]

                def getIndexData(self, inds, params, dtype):
                    return gbs[int(np.asarray(inds).reshape(-1)[0]) - 1]
            out = host.generateAnswers(GenLoader(), 'val', dict(beamSize=bs, beamLen=bl, maxThreads=2))
            for c in range(2):
                want_txt = []
                for toks in z['gen.tokens'][c]:
                    s_ = ''
                    for t in toks:
                        if t > 0:
                            s_ += ' ' + words[int(t)]
                            if words[int(t)] == '<END>':
                                break
                    want_txt.append(s_)
                assert [d['answer'] for d in out[c]['dialog']] == want_txt, (type(host).__name__, c)
