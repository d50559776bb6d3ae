"""Fixtures made by EXECUTING the reference's own Lua (model.lua, encoders/*.lua, decoders/*.lua, model_utils/*.lua, utils.lua) in
this container -- tests/golden/make_reference_goldens.py: the reference's sources on tests/luavm (Lua 5.1 evaluator + Torch7 tensor
stub) over tests/luavm/nn7.py (numpy restatement of the nn / nngraph / rnn modules the reference composes; those packages are
third-party and absent from /root/reference).  tests/golden/ref__<encoder>__<decoder>.npz hold, per pair: the parameters the
reference initialised (rounded to fp32), one synthetic batch, and what the reference computed from them --

  eval.*    Model:forwardBackward in evaluate() mode: loss + every gradient tensor (fp64)
  train.*   the same in training() mode, with the noise every nn.Dropout drew recorded under mask.<site>
  step.*    Model:trainIteration: the Adam update of every tensor, runningLoss (the global), the decayed learning rate
  rank.*    Model:retrieveBatch -> utils.computeRanks: the decoder's scores, the GT ranks and all ranks (disc); for gen the 100
            candidates of rbatch.* scored by forwardConnect + decoder forward + utils.computeLhood (model.lua:392-420)
  beam.*    Model:generateAnswers, beam search (model.lua:432-573), one dialog x 10 rounds: the winning token vectors         (gen)
  multi.*   two MORE Model:trainIteration calls on fresh batches (Adam at t = 2, 3, lr decay, runningLoss moving average)   (three pairs)
  split.*   Model:evaluate / retrieve (+ utils.processRanks) / predict (model.lua:109-246) over a 3-dialog split in two batches:
            the printed loss / perplexity / R@k / MRR, the ground-truth ranks and all ranks          (three pairs)

CPU: oracle/visdial_oracle.py reproduces all of it (this is what pins the oracle to the reference).
GPU: the HIP path, fed the same parameters / batch / masks through the C ABI, matches within the fp32 tolerance (1e-4).
Nothing here reads /root/reference."""
import glob
import json
import os

import numpy as np
import pytest
import torch

from conftest import ROOT, grad_mismatches, unexplained_rank_flips
from oracle import visdial_oracle as vo
from visdial_amd.opts import derive

FILES = sorted(glob.glob(os.path.join(ROOT, 'tests', 'golden', 'ref__*__*.npz')))
IDS = [os.path.basename(f)[5:-4] for f in FILES]


def load(path):
    z = np.load(path)
    get = lambda pre: {k[len(pre):]: z[k] for k in z.files if k.startswith(pre)}
    p = derive(json.loads(str(z['opt.json'])))
    return p['encoder'], p['decoder'], p, z, get('param.'), get('batch.'), get('mask.')


def test_fixture_set_is_complete():
    assert len(FILES) == 14
    encs = {i.split('__')[0] for i in IDS}
    assert len(encs) == 11 and sum(i.endswith('__gen') for i in IDS) == 3          # every encoder with disc, three with gen


@pytest.mark.parametrize("path", FILES, ids=IDS)
def test_oracle_reproduces_the_executed_reference(path):
    enc, dec, p, z, P, batch, masks = load(path)
    get = lambda pre: {k[len(pre):]: z[k] for k in z.files if k.startswith(pre)}
    P64 = {k: v.astype(np.float64) for k, v in P.items()}
    assert [n for n, _, _ in vo.param_spec(enc, dec, p)] and set(P64) == {n for n, _, _ in vo.param_spec(enc, dec, p)}
    for n, shape, _ in vo.param_spec(enc, dec, p):                      # the reference's modules have the shapes the oracle declares
        assert P64[n].shape == tuple(shape), n

    # evaluate(): loss and every gradient, fp64 against fp64
    r = vo.forward_backward(enc, dec, P64, p, batch, None)
    assert abs(r['loss'] - float(z['eval.loss'])) < 1e-10 * max(1.0, abs(float(z['eval.loss'])))
    assert not grad_mismatches(r['grads'], get('eval.grad.'), tol=1e-9)

    # training(): with the noise the reference's Dropout nodes drew (gradients stored in fp32)
    drop = {k: v.astype(np.float64) for k, v in masks.items()} if masks else None
    rt = vo.forward_backward(enc, dec, P64, p, batch, drop)
    assert abs(rt['loss'] - float(z['train.loss'])) < 1e-10 * max(1.0, abs(float(z['train.loss'])))
    assert not grad_mismatches(rt['grads'], get('train.grad.'), tol=1e-6)
    if masks:
        assert abs(float(z['train.loss']) - float(z['eval.loss'])) > 1e-6          # (the masks did something)

    # trainIteration (model.lua:66-106): clamp + adam + runningLoss + lr decay
    delta = get('step.delta.')
    for k in P64:
        w2, _ = vo.clamp_adam(P64[k].reshape(-1), r['grads'][k].reshape(-1), {}, p['learningRate'])
        assert np.abs((w2 - P64[k].reshape(-1)) - delta[k].reshape(-1)).max() < 1e-9, k
    want_rl = r['loss'] / max(int((batch['answer_out'] > 0).sum()), 1) if dec == 'gen' else r['loss']
    assert abs(float(z['step.runningLoss']) - want_rl) < 1e-10 * max(1.0, abs(want_rl))
    assert abs(float(z['step.learningRate']) - p['learningRate'] * p['lrDecayRate']) < 1e-15

    # retrieveBatch + utils.computeRanks on the updated parameters
    if dec == 'disc':
        P1 = {k: P64[k] + delta[k].astype(np.float64) for k in P64}
        ev = vo.forward_backward(enc, dec, P1, p, batch, None, only_forward=True)
        assert np.abs(ev['scores'] - z['rank.scores']).max() < 1e-8
        np.testing.assert_array_equal(vo.compute_ranks(z['rank.scores']), z['rank.all'])          # the rule itself, exactly
        np.testing.assert_array_equal(vo.compute_ranks(z['rank.scores'], batch['answer_ind'].reshape(-1) - 1), z['rank.gt'])
        flipped, bad = unexplained_rank_flips(ev['scores'], z['rank.scores'], tol=1e-6)           # on the oracle's own scores: up to ties
        assert not bad, bad[:5]
    else:
        P1 = {k: P64[k] + delta[k].astype(np.float64) for k in P64}
        rb = dict(batch, option_in=z['rbatch.option_in'], option_out=z['rbatch.option_out'], answer_ind=z['rbatch.answer_ind'])
        scores = vo.retrieve(enc, dec, P1, p, rb)
        assert np.abs(scores - z['rank.scores']).max() < 1e-8
        np.testing.assert_array_equal(vo.compute_ranks(z['rank.scores']), z['rank.all'])             # what the reference returned
        np.testing.assert_array_equal(vo.compute_ranks(z['rank.scores'], rb['answer_ind'].reshape(-1) - 1), z['rank.gt'])
        # generateAnswers: the reference's winning beams, token for token (incl. the never-filled-slot corner: a beam whose last
        # token is 0 gets an all-zero log-probability row from MaskZero, decoders/gen.lua:23-24)
        bs, bl, START, END = (int(v) for v in z['beam.params'])
        Pb = dict(P1)
        Pb['vocab.b'] = z['beam.vocab_b'].astype(np.float64)
        gb = {k: v[:1] for k, v in batch.items()}
        got = vo.generate_beam(enc, Pb, p, gb, bs, bl, START, END)
        np.testing.assert_array_equal(np.array([t for t, _ in got]), z['beam.tokens'])
        np.testing.assert_allclose([sc for _, sc in got], z['beam.scores'], atol=1e-9)
        assert (z['beam.tokens'] == END).any(axis=1).all()                        # every round finished a beam


@pytest.mark.gpu
@pytest.mark.parametrize("path", FILES, ids=IDS)
def test_hip_path_matches_the_executed_reference(path):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from visdial_amd.model import Model
    enc, dec, p, z, P, batch, masks = load(path)
    get = lambda pre: {k[len(pre):]: z[k] for k in z.files if k.startswith(pre)}
    model = Model(p)
    model.set_parameters_dict(P)

    model.wrapper.evaluate()
    model.wrapper.zeroGradParameters()
    loss = model.forwardBackward(batch)
    ref = float(z['eval.loss'])
    assert abs(loss - ref) < 1e-4 * max(1.0, abs(ref))
    bad = grad_mismatches(model.get_gradients_dict(), get('eval.grad.'))
    assert not bad, bad

    if masks:
        model.wrapper.training()
        model.set_dropout_masks(masks)
        model.wrapper.zeroGradParameters()
        loss = model.forwardBackward(batch)
        ref = float(z['train.loss'])
        assert abs(loss - ref) < 1e-4 * max(1.0, abs(ref))
        bad = grad_mismatches(model.get_gradients_dict(), get('train.grad.'))
        assert not bad, bad
        model.set_dropout_masks(None)

    # Model:trainIteration on a dataloader that serves this batch, dropout off like the fixture's step
    class OneBatch(object):
        def getTrainBatch(self, params):
            return batch
    model.wrapper.evaluate()
    model.runningLoss = 0
    model.trainIteration(OneBatch())
    W1 = model.get_parameters_dict()
    G1 = model.get_gradients_dict()          # (the fused clamp + adam kernel leaves the CLAMPED gradient it used in wrapperdW)
    assert not grad_mismatches(G1, {k: np.clip(v, -5.0, 5.0) for k, v in get('eval.grad.').items()})
    if dec == 'gen':
        assert max(np.abs(v).max() for v in get('eval.grad.').values()) > 5.0          # (the summed gen loss does reach the clamp)
    delta = get('step.delta.')
    eps1 = 1e-8 / np.sqrt(1 - 0.999)
    for k in P:
        got = (W1[k].astype(np.float64) - P[k].astype(np.float64)).reshape(-1)
        ulp = 2e-7 * max(float(np.abs(P[k]).max()), p['learningRate'])
        # (1) the update rule itself, element by element on the DEVICE's own gradient: clamp + adam of model.lua:96-99 /
        #     optim_updates.lua:62-91 in fp32
        w2, _ = vo.clamp_adam(P[k].astype(np.float64).reshape(-1), G1[k].astype(np.float64).reshape(-1), {}, p['learningRate'])
        assert np.abs(got - (w2 - P[k].astype(np.float64).reshape(-1))).max(initial=0.0) < 1e-5 * p['learningRate'] + ulp, k
        # (2) against the reference's step.  Adam's first step is lr * g / (|g| + eps'), eps' = 1e-8 / sqrt(1 - beta2) = 3.2e-7: a
        #     gradient that is off by dg moves it by lr * eps' * dg / (|g| + eps')^2 -- nothing for |g| >> eps', everything for
        #     |g| ~ eps'.  dg: 1e-3 of the tensor's largest element (fp32 accumulation over thousands of cancelling rows; the
        #     tensor-level bound on the gradient is the 1e-4 relative L2 checked above), at least the 1e-6 absolute floor that
        #     grad_mismatches grants a tensor whose exact gradient is zero (att.b, att_q.b: a bias in front of a softmax -- its fp32
        #     gradient is rounding noise around eps', so its first Adam step is noise too, here and in the reference's own fp32).
        g = np.abs(z['eval.grad.' + k].reshape(-1))
        tol = p['learningRate'] * (eps1 * max(1e-3 * g.max(initial=0.0), 1e-6) / (g + eps1) ** 2 + 1e-4) + ulp
        worst = np.abs(got - delta[k].reshape(-1)) / tol
        assert worst.max(initial=0.0) < 1.0, (k, float(worst.max()), int(worst.argmax()))
    want_rl = float(z['step.runningLoss'])
    assert abs(model.runningLoss - want_rl) < 1e-4 * max(1.0, abs(want_rl))
    assert abs(model.optims['learningRate'] - float(z['step.learningRate'])) < 1e-12

    if dec == 'disc':
        # ranks from the fixture's UPDATED parameters (not this run's, whose near-zero-gradient elements may have stepped differently)
        model.set_parameters_dict({k: (P[k].astype(np.float64) + delta[k]).astype(np.float32) for k in P})
        model.wrapper.evaluate()
        p['useGt'] = False
        ranks = np.asarray(model.retrieveBatch(batch))
        dev_scores = model.decoder.output.cpu().numpy()
        assert np.abs(dev_scores - z['rank.scores']).max() < 1e-4
        np.testing.assert_array_equal(ranks.reshape(dev_scores.shape), vo.compute_ranks(dev_scores))
        flipped, bad = unexplained_rank_flips(dev_scores, z['rank.scores'])
        assert not bad, bad[:10]
        assert (ranks.reshape(dev_scores.shape) != z['rank.all']).sum() <= 2 * flipped
        p['useGt'] = True
        gt = np.asarray(model.retrieveBatch(batch)).reshape(-1)
        assert np.abs(gt - z['rank.gt']).sum() <= flipped
    else:
        # gen retrieval (model.lua:392-420) and beam search (model.lua:432-573) from the fixture's updated parameters, both hosts
        from visdial_amd.native import NativeModel
        P1 = {k: (P[k].astype(np.float64) + delta[k]).astype(np.float32) for k in P}
        rb = dict(batch, option_in=z['rbatch.option_in'], option_out=z['rbatch.option_out'], answer_ind=z['rbatch.answer_ind'])
        bs, bl, START, END = (int(v) for v in z['beam.params'])
        words = {i: 'w%d' % i for i in range(1, p['vocabSize'] + 1)}
        words[START], words[END] = '<START>', '<END>'
        gb = {k: v[:1] for k, v in batch.items()}

        class GenLoader(object):
            word2ind, ind2word, numThreads, unique_img_val = {'<START>': START, '<END>': END}, words, {'val': 1}, [4711]

            def getIndexData(self, inds, params, dtype):
                return gb
        want_txt = []
        for toks in z['beam.tokens']:
            sent = ''
            for t in toks:
                if t > 0:
                    sent += ' ' + words[int(t)]
                    if words[int(t)] == '<END>':
                        break
            want_txt.append(sent)
        for host in (model, NativeModel(p)):
            host.set_parameters_dict(P1)
            (host.wrapper.evaluate if hasattr(host, 'wrapper') else (lambda: host.training(False)))()
            p['useGt'] = False
            ranks = np.asarray(host.retrieveBatch(rb)).reshape(z['rank.all'].shape)
            dev = (host.scores.cpu().numpy() if hasattr(host, 'wrapper') else host.scores(*z['rank.all'].shape)).astype(np.float64)
            p['useGt'] = True
            gt = np.asarray(host.retrieveBatch(rb)).reshape(-1)
            assert np.abs(dev - z['rank.scores']).max() < 1e-4 * max(1.0, np.abs(z['rank.scores']).max())
            np.testing.assert_array_equal(ranks, vo.compute_ranks(dev))
            flipped, bad = unexplained_rank_flips(dev, z['rank.scores'])
            assert not bad, bad[:10]
            assert (ranks != z['rank.all']).sum() <= 2 * flipped and np.abs(gt - z['rank.gt']).sum() <= flipped
            Pb = dict(P1)
            Pb['vocab.b'] = z['beam.vocab_b'].astype(np.float32)
            host.set_parameters_dict(Pb)
            out = host.generateAnswers(GenLoader(), 'val', dict(beamSize=bs, beamLen=bl, maxThreads=1))
            assert out[0]['image_id'] == 4711
            assert [d['answer'] for d in out[0]['dialog']] == want_txt, (type(host).__name__, [d['answer'] for d in out[0]['dialog']], want_txt)


# ---------------------------------------------------------------------------------------------------------------------- split loops
SPLIT = [f for f in FILES if 'split.loss' in np.load(f).files]


class _SplitLoader(object):
    """what Model:evaluate / retrieve / predict use of a dataloader: the fixture's two batches (2 + 1 dialogs)"""
    numThreads, unique_img_val, val_num_rounds = {'val': 3}, [11, 12, 13], [10, 9, 10]

    def __init__(self, batches):
        self.batches = batches

    def getTestBatch(self, start, params, dtype):
        return (self.batches[0], 3) if start == 1 else (self.batches[1], 4)


def _split_fixture(path):
    z = np.load(path)
    p = derive(json.loads(str(z['opt.json'])))
    get = lambda pre: {k[len(pre):]: z[k] for k in z.files if k.startswith(pre)}
    return z, p, get('split.param.'), [get('split.batch0.'), get('split.batch1.')]


def _check_split(host, z, exact):
    """run the product's three loops (visdial_amd/split_eval.py) on `host` and compare with what the reference printed / returned"""
    D = _SplitLoader(host._batches)
    loss, ppl = host.evaluate(D, 'val')
    tol = 1e-6 if exact else 1e-4
    assert abs(loss - float(z['split.loss'])) < tol * max(1.0, abs(float(z['split.loss']))) + 1e-6          # (%f: six decimals were printed)
    assert abs(ppl - float(z['split.ppl'])) < (2e-6 if exact else 2e-4) * float(z['split.ppl']) + 1e-6
    metrics, recs = host.retrieve(D, 'val')
    gt = z['split.gt_ranks']
    rounds = [10, 9, 10]
    flips = 0
    assert [(r['image_id'], r['round_id']) for r in recs] == [(11 + i, j + 1) for i in range(3) for j in range(rounds[i])]
    got = np.array([r['ranks'] for r in recs])
    want = np.array([gt[i, j] for i in range(3) for j in range(rounds[i])], dtype=np.float64)
    if exact:
        np.testing.assert_array_equal(got, want)                  # (the ranks themselves: the reference's GPU path; its CPU path returns 1 / rank)
        np.testing.assert_allclose([metrics[k] for k in ('r@1', 'r@5', 'r@10', 'medianR', 'meanR', 'meanRR')], z['split.metrics'], atol=1e-6)
    else:
        flips = int(np.abs(got - want).sum())
        assert (got != want).mean() < 0.1 and np.abs(got - want).max() <= 2              # fp32 near ties only
        assert abs(metrics['meanR'] - z['split.metrics'][4]) <= flips / 30.0 + 1e-6
    pred = host.predict(D, 'val')
    allr = z['split.all_ranks']
    assert [(r['image_id'], r['round_id']) for r in pred] == [(11 + i, j + 1) for i in range(3) for j in range(rounds[i])]
    got = np.array([r['ranks'] for r in pred])
    want = np.array([allr[i, j] for i in range(3) for j in range(rounds[i])], dtype=np.float64)
    if exact:
        np.testing.assert_array_equal(got, want)
    else:
        assert (got != want).mean() < 0.02


def test_split_fixtures_present():
    assert len(SPLIT) == 3


@pytest.mark.parametrize("path", SPLIT, ids=[os.path.basename(f)[5:-4] for f in SPLIT])
def test_split_loops_on_an_oracle_host_reproduce_the_executed_reference(path):
    """visdial_amd/split_eval.py (the loops both product hosts share) driven by the fp64 oracle instead of the device"""
    from visdial_amd.split_eval import SplitEval
    z, p, P, batches = _split_fixture(path)
    P64 = {k: v.astype(np.float64) for k, v in P.items()}
    enc, dec = p['encoder'], p['decoder']

    class OracleHost(SplitEval):
        params, _batches = p, batches

        def _set_training(self, on):
            pass

        def forwardBackward(self, batch, onlyForward=False):
            return vo.forward_backward(enc, dec, P64, p, batch, None, only_forward=True)['loss']

        def retrieveBatch(self, batch):
            sc = vo.retrieve(enc, dec, P64, p, batch)
            return vo.compute_ranks(sc, batch['answer_ind'].reshape(-1) - 1) if self.params['useGt'] else vo.compute_ranks(sc)
    _check_split(OracleHost(), z, exact=True)


@pytest.mark.gpu
@pytest.mark.parametrize("path", SPLIT, ids=[os.path.basename(f)[5:-4] for f in SPLIT])
def test_split_loops_on_the_hip_hosts_match_the_executed_reference(path):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from visdial_amd.model import Model
    from visdial_amd.native import NativeModel
    z, p, P, batches = _split_fixture(path)
    for host in (Model(dict(p)), NativeModel(dict(p))):
        host.set_parameters_dict(P)
        host._batches = batches
        _check_split(host, z, exact=False)


# ---------------------------------------------------------------------------------------------------------------------- three iterations
@pytest.mark.parametrize("path", SPLIT, ids=[os.path.basename(f)[5:-4] for f in SPLIT])
def test_oracle_reproduces_three_reference_iterations(path):
    enc, dec, p, z, P, batch, masks = load(path)
    get = lambda pre: {k[len(pre):]: z[k] for k in z.files if k.startswith(pre)}
    state = {}
    vo.train_iteration(enc, dec, {k: v.astype(np.float64) for k, v in P.items()}, p, batch, None, state, p['learningRate'])
    Pc = {k: v.astype(np.float64) for k, v in get('multi.param.').items()}
    start = {k: v.copy() for k, v in Pc.items()}
    lr, rl = (float(v) for v in z['multi.start'])
    for it in range(2):
        b = get('multi.batch%d.' % it)
        Pc, r = vo.train_iteration(enc, dec, Pc, p, b, None, state, lr)
        if lr > p['minLRate']:
            lr *= p['lrDecayRate']
        cur = r['loss'] / max(int((b['answer_out'] > 0).sum()), 1) if dec == 'gen' else r['loss']
        rl = 0.95 * rl + 0.05 * cur if rl > 0 else cur
    delta = get('multi.delta.')
    for k in Pc:
        assert np.abs((Pc[k] - start[k]) - delta[k]).max() < 1e-9, k                 # (deltas stored in fp32: |delta| ~ 3e-3)
    assert abs(lr - float(z['multi.end'][0])) < 1e-15 and abs(rl - float(z['multi.end'][1])) < 1e-9 * max(1.0, abs(rl))


@pytest.mark.gpu
@pytest.mark.parametrize("path", SPLIT, ids=[os.path.basename(f)[5:-4] for f in SPLIT])
def test_hip_hosts_reproduce_three_reference_iterations(path):
    """Adam's moments / bias correction at t = 2, 3, the lr decay and the runningLoss moving average across calls, through both hosts"""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from visdial_amd.model import Model
    from visdial_amd.native import NativeModel
    enc, dec, p, z, P, batch, masks = load(path)
    get = lambda pre: {k[len(pre):]: z[k] for k in z.files if k.startswith(pre)}
    batches = [batch, get('multi.batch0.'), get('multi.batch1.')]
    delta = get('multi.delta.')
    start = get('multi.param.')

    class Loader(object):
        def __init__(self):
            self.i = 0

        def getTrainBatch(self, params, **_k):
            b = batches[min(self.i, 2)]
            self.i += 1
            return b
    for host in (Model(dict(p)), NativeModel(dict(p))):
        host.set_parameters_dict(P)
        (host.wrapper.evaluate if hasattr(host, 'wrapper') else (lambda: host.training(False)))()
        dl = Loader()
        host.runningLoss = 0
        host.trainIteration(dl)                                   # t = 1 on the first batch (checked above); then the fixture's edited parameters
        if hasattr(host, 'synchronize'):
            host.synchronize()
        host.set_parameters_dict(start)
        host.optims['learningRate'], host.runningLoss = float(z['multi.start'][0]), float(z['multi.start'][1])
        host.trainIteration(dl)                                   # (the pipelined hosts have already prefetched batches[1]: inputs only)
        host.trainIteration(dl)
        W = host.get_parameters_dict()
        eps1 = 1e-8 / np.sqrt(1 - 0.999)
        for k in start:
            got = (W[k].astype(np.float64) - start[k].astype(np.float64)).reshape(-1)
            # two Adam steps of <= lr each; an element's step is sensitive to fp32 gradient noise only where |g| ~ eps' (see above):
            # bound the worst element by 2 lr for those, and the bulk by 1e-3 of the step
            err = np.abs(got - delta[k].reshape(-1))
            assert err.max(initial=0.0) <= 2.0001 * p['learningRate'] + 4e-7 * np.abs(start[k]).max(), k
            if np.abs(delta[k]).max() < 1e-9:          # the reference did not move it (att.b: exact gradient 0): the fp32 step is rounding noise
                continue
            assert np.median(err) < 2e-6 * p['learningRate'] + 4e-7 * max(float(np.abs(start[k]).max()), 1e-3), (k, float(np.median(err)))
            assert np.linalg.norm(err) <= 0.02 * np.linalg.norm(delta[k]) + 1e-6, (k, float(np.linalg.norm(err)), float(np.linalg.norm(delta[k])))
        assert abs(host.optims['learningRate'] - float(z['multi.end'][0])) < 1e-7 * p['learningRate']      # (the library keeps it in fp32)
        assert abs(host.runningLoss - float(z['multi.end'][1])) < 1e-4 * max(1.0, abs(float(z['multi.end'][1])))
