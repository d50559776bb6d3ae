#!/usr/bin/env python
"""EXECUTES THE REFERENCE'S OWN LUA (model.lua, encoders/*.lua, decoders/*.lua, model_utils/*.lua, utils.lua) inside tests/luavm and
records what it computes, as fixtures under tests/golden/ref__<encoder>__<decoder>.npz.  Runs in the BUILD CONTAINER only
(/root/reference does not travel); the fixtures and this script are what is committed.

What runs: the reference's files, unmodified, read in place from /root/reference: class Model (model.lua:8-106: construction through
the plug-in files, wrapper / getParameters, forwardBackward, trainIteration with clamp + optim_updates.lua's adam, retrieveBatch +
utils.computeRanks).  What stands in for Torch7: tests/luavm (the Lua evaluator), tests/luavm/torch7.py (tensors) and
tests/luavm/nn7.py (nn / nngraph / rnn module semantics restated from their published behaviour, independent of the oracle).

What is recorded per pair:
  * the parameters the reference's constructors drew (by module), the batch, the Dropout keep-masks its forward pass drew;
  * loss and every gradient tensor of Model:forwardBackward in evaluate() and in training() mode;
  * the parameters after one Model:trainIteration (zeroGradParameters, forwardBackward, clamp(-5, 5), adam, lr decay) + runningLoss + lr;
  * the GT ranks / all ranks Model:retrieveBatch produces through the reference's utils.computeRanks;
  * the ORDER of the tensors in `wrapper:getParameters()` -- for the four nngraph encoders this is nngraph's forward-node order, which
    visdial_amd/t7.py needs to load reference checkpoints (tests/golden/reference_param_order.json).
The oracle is NOT an input of this script: it is compared with these fixtures by tests/test_reference_goldens.py, and below (a
cross-check at generation time, so that a disagreement is seen before a fixture is written).
"""
import json
import os
import re
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
REF = '/root/reference'

from conftest import small_params  # noqa: E402
from luavm import new_vm, to_lua, to_py  # noqa: E402
from luavm import nn7  # noqa: E402
from luavm.interp import LuaTable, call, first, index  # noqa: E402
from oracle import visdial_oracle as vo  # noqa: E402
from visdial_amd.dataloader import SyntheticDataloader  # noqa: E402
from visdial_amd.opts import derive  # noqa: E402

ALL_ENC = ['lf-ques', 'lf-ques-im', 'lf-ques-hist', 'lf-ques-im-hist', 'lf-att-ques-im-hist', 'hre-ques-hist',
           'hre-ques-im-hist', 'hrea-ques-im-hist', 'mn-ques-hist', 'mn-ques-im-hist', 'mn-att-ques-im-hist']
PAIRS = [(e, 'disc') for e in ALL_ENC] + [('lf-ques', 'gen'), ('lf-ques-im-hist', 'gen'), ('mn-att-ques-im-hist', 'gen')]
# Dropout nodes in the order the encoder files CONSTRUCT them -> the site names of this repo (visdial_amd/dataloader.py)
DROP_SITES = {'mn-att-ques-im-hist': ['q_emb', 'h_emb', 'hatt', 'img_tr', 'iqc', 'u'], 'mn-ques-hist': ['q_emb', 'h_emb', 'hatt'],
              'mn-ques-im-hist': ['q_emb', 'h_emb', 'hatt'], 'lf-att-ques-im-hist': ['q_emb', 'h_emb', 'img_tr', 'iqc', 'u'],
              'hrea-ques-im-hist': ['img'], 'hre-ques-hist': [], 'hre-ques-im-hist': []}
# module names in the order the encoder FILE constructs them, where that differs from this repo's declaration order
# (encoders/hre-ques-hist.lua:12-36 builds the question LSTMs before the history LSTMs)
CREATION_ORDER = {'hre-ques-hist': ['embed', 'ques*', 'hist*', 'dialog']}
BATCH_TYPES = {'ques_fwd': 'Long', 'hist': 'Long', 'options': 'Long', 'answer_in': 'Long', 'answer_out': 'Long', 'answer_ind': 'Long',
               'img_feat': 'Double', 'option_in': 'Long', 'option_out': 'Long'}


class Capture(object):
    """the VM's stdout: everything is dropped except while .on is set"""

    def __init__(self):
        self.on, self.text = False, ''

    def write(self, s):
        if self.on:
            self.text += s

    def flush(self):
        pass


def reference_vm(seed):
    out = Capture()
    vm = new_vm(search=[REF, os.path.join(ROOT, 'tests', 'lua_ref_stubs')], stdout=out)
    vm.captured = out
    NN = nn7.install(vm, seed=seed)
    cj = LuaTable()
    cj.set('decode', lambda s, *_: to_lua(vm, json.loads(s)))
    cj.set('encode', lambda t, *_: json.dumps(to_py(t)))
    vm.preload.set('cjson', lambda *_: cj)
    vm.preload.set('hdf5', lambda *_: LuaTable())
    vm.globals.set('runningLoss', 0)
    vm.dostring("require 'model'")
    return vm, NN


def param_modules(mod, out=None, seen=None):
    """the DISTINCT parameterised nn7 modules under `mod` (shared clones once, by storage), with their creation index"""
    out, seen = ([] if out is None else out), (set() if seen is None else seen)
    ps = mod.params()
    key = ps[0][0].a.__array_interface__['data'][0] if ps else None       # (after getParameters() every tensor is a view of ONE storage)
    if ps and key not in seen:
        seen.add(key)
        out.append(mod)
    for c in mod.children():
        if isinstance(c, nn7.Module):
            param_modules(c, out, seen)
    return out


def dropouts(mod, out=None):
    out = [] if out is None else out
    if isinstance(mod, nn7.Dropout):
        out.append(mod)
    for c in mod.children():
        if isinstance(c, nn7.Module):
            dropouts(c, out)
    return out


def run_pair(enc, dec, seed=7, extra=None, order_only=False):
    # maxQuesCount = 10 and 100 options are hard-coded in the reference (model.lua:281, decoders/disc.lua:10); everything else small.
    # E = 24 where Dropout(0.5) follows the embedding: a row it zeroes ENTIRELY reads as padding to SeqLSTM:maskZero() (2^-24 per row here,
    # 2^-300 at the real E = 300) -- the one place where "mask by token id" (this repo) and "mask by zero vector" (rnn) could differ
    graph = enc.startswith('mn') or enc.startswith('lf-att')
    p = derive(small_params(encoder=enc, decoder=dec, maxQuesCount=10, numOptions=100, batchSize=2, vocabSize=30, rnnHiddenSize=32,
                            commonEmbeddingSize=12, maxQuesLen=5, maxHistoryLenPerRound=7, maxAnsLen=4, imgSpatialSize=3, imgFeatureSize=8,
                            imgEmbedSize=8, embedSize=24 if graph else 12, **(extra or {})))         # (sizes the HIP path accepts: H % 32, E % 4, ...)
    vm, NN = reference_vm(seed)
    lp = dict({k: v for k, v in p.items() if isinstance(v, (int, float, str, bool))}, gpuid=-1, weightInit='xavier', ansHiddenSize=0)
    model = first(call(vm.globals.get('Model'), [to_lua(vm, lp)]))
    wrapper = index(model, 'wrapper')
    inv = lambda obj, name, *a: call(index(obj, name), [obj] + list(a))

    # ---- names: the parameterised modules in the order the Lua files construct them == this repo's declaration order
    spec = vo.param_spec(enc, dec, p)
    groups, cur = [], None
    for name, shape, _ in spec:
        base = name.rsplit('.', 1)[0] if '.' in name else name
        if base != cur:
            groups.append([])
            cur = base
        groups[-1].append((name, tuple(shape)))
    if enc in CREATION_ORDER:       # this file constructs its modules in another order than this repo declares them
        by = {g[0][0].rsplit('.', 1)[0] if '.' in g[0][0] else g[0][0]: g for g in groups}
        names = [n for t in CREATION_ORDER[enc] for n in ([t] if not t.endswith('*') else ['%s%d' % (t[:-1], l + 1) for l in range(p['numLayers'])])]
        dec_groups = [g for g in groups if (g[0][0].split('.')[0] if '.' in g[0][0] else g[0][0]) not in names]
        groups = [by[n] for n in names] + dec_groups
    mods = sorted(param_modules(wrapper), key=lambda m: m.created)
    assert len(mods) == len(groups), (enc, dec, len(mods), len(groups))
    flat_W, flat_dW = index(model, 'wrapperW'), index(model, 'wrapperdW')
    base = flat_W.st.base.__array_interface__['data'][0]
    order = []
    for m, g in zip(mods, groups):
        ps = m.params()
        assert len(ps) == len(g), (enc, [x[0] for x in g])
        for (w, dw), (name, shape) in zip(ps, g):
            assert tuple(w.a.shape) == shape, (enc, dec, name, w.a.shape, shape)
            order.append(((w.a.__array_interface__['data'][0] - base) // 8, w, name))      # where getParameters() put it
    order.sort(key=lambda e: e[0])
    off = 0
    for o, w, name in order:
        assert o == off, (enc, dec, name, o, off)           # back to back, every tensor once
        off += w.a.size
    assert off == flat_W.a.size
    order = [(w, name) for _, w, name in order]
    names_in_flat_order = [n for _, n in order]

    def named(flat):
        out, o = {}, 0
        for w, n in order:
            out[n] = np.array(flat.a[o:o + w.a.size].reshape(w.a.shape), dtype=np.float64)
            o += w.a.size
        return out

    flat_W.a[...] = flat_W.a.astype(np.float32).astype(np.float64)      # fp32-representable parameters: the HIP path gets exactly the same ones
    if order_only:
        return p, None, names_in_flat_order, []
    P = named(flat_W)
    dl = SyntheticDataloader(p, seed=11)
    batch = dl.getTrainBatch(p)
    lua_batch = lambda: to_lua(vm, {k: v for k, v in batch.items() if isinstance(v, np.ndarray)}, BATCH_TYPES)
    rec = {}
    rec.update({'batch.' + k: v for k, v in batch.items() if isinstance(v, np.ndarray)})
    rec['opt.json'] = np.array(json.dumps({k: v for k, v in p.items() if isinstance(v, (int, float, str, bool))}))
    rec['order.json'] = np.array(json.dumps(names_in_flat_order))

    # ---- evaluate(): dropout off
    inv(wrapper, 'evaluate')
    inv(wrapper, 'zeroGradParameters')
    loss_ev = first(inv(model, 'forwardBackward', lua_batch()))
    g_ev = named(flat_dW)
    P = named(flat_W)                 # (nn.LookupTableMaskZero re-zeroes its pad row on every forward: the parameters as the step saw them)
    rec.update({'param.' + k: v.astype(np.float32) for k, v in P.items()})
    assert all((rec['param.' + k].astype(np.float64) == v).all() for k, v in P.items())
    ref = vo.forward_backward(enc, dec, P, p, batch, None)
    worst = max(float(np.linalg.norm(g_ev[k] - ref['grads'][k]) / max(np.linalg.norm(ref['grads'][k]), 1e-30)) for k in g_ev
                if np.abs(ref['grads'][k]).max() > 1e-12)
    assert abs(loss_ev - ref['loss']) < 1e-9 * max(1, abs(ref['loss'])) and worst < 1e-8, (enc, dec, 'evaluate', loss_ev, ref['loss'], worst)
    rec['eval.loss'] = np.float64(loss_ev)
    rec.update({'eval.grad.' + k: v for k, v in g_ev.items()})
    report = ['evaluate: |dloss| %.1e worst grad %.1e' % (abs(loss_ev - ref['loss']), worst)]

    # ---- training(): the Dropout nodes draw their own noise; record it by site
    inv(wrapper, 'training')
    inv(wrapper, 'zeroGradParameters')
    NN.dropout_log = []
    loss_tr = first(inv(model, 'forwardBackward', lua_batch()))
    g_tr = named(flat_dW)
    drops = sorted(dropouts(wrapper), key=lambda m: m.created)
    sites = DROP_SITES.get(enc, ['fuse'] if p.get('dropout', 0.5) > 0 else [])
    assert len(drops) == len(sites), (enc, len(drops), sites)
    masks = {s: d.noise for s, d in zip(sites, drops) if d.noise is not None}
    ref_t = vo.forward_backward(enc, dec, P, p, batch, {k: v.astype(np.float64) for k, v in masks.items()} if masks else None)
    worst_t = max(float(np.linalg.norm(g_tr[k] - ref_t['grads'][k]) / max(np.linalg.norm(ref_t['grads'][k]), 1e-30)) for k in g_tr
                  if np.abs(ref_t['grads'][k]).max() > 1e-12)
    assert abs(loss_tr - ref_t['loss']) < 1e-9 * max(1, abs(ref_t['loss'])) and worst_t < 1e-8, (enc, dec, 'training', loss_tr, ref_t['loss'], worst_t)
    rec['train.loss'] = np.float64(loss_tr)
    rec.update({'train.grad.' + k: v.astype(np.float32) for k, v in g_tr.items()})         # (fp32 storage; the 1e-15 agreement is in the log)
    rec.update({'mask.' + k: v.astype(np.uint8) for k, v in masks.items()})
    report.append('training (%d dropout sites): |dloss| %.1e worst grad %.1e' % (len(masks), abs(loss_tr - ref_t['loss']), worst_t))

    # ---- Model:trainIteration (model.lua:66-106) on a dataloader that serves this batch; dropout off so the step is reproducible
    inv(wrapper, 'evaluate')

    class OneBatch(object):
        lua_type = 'table'

        def lua_index(self, k):
            if k == 'getTrainBatch':
                return lambda *_a: lua_batch()
            return None
    orig_training = index(wrapper, 'training')
    vm.globals.set('runningLoss', 0)
    inv(model, 'trainIteration', OneBatch())
    W1 = named(flat_W)
    rec.update({'step.delta.' + k: (W1[k] - P[k]).astype(np.float32) for k in W1})        # the Adam step itself (|delta| ~ lr: 1e-10 absolute in fp32)
    rec['step.runningLoss'] = np.float64(vm.globals.get('runningLoss'))
    rec['step.learningRate'] = np.float64(index(index(model, 'optims'), 'learningRate'))
    worst_w = 0.0
    for k in P:
        w2, _ = vo.clamp_adam(P[k].reshape(-1), g_ev[k].reshape(-1), {}, p['learningRate'])
        worst_w = max(worst_w, float(np.abs(W1[k].reshape(-1) - w2).max()))
    assert worst_w < 1e-12, (enc, dec, 'adam', worst_w)
    want_rl = ref['loss'] / max(int((batch['answer_out'] > 0).sum()), 1) if dec == 'gen' else ref['loss']
    assert abs(rec['step.runningLoss'] - want_rl) < 1e-9 * max(1, abs(want_rl))
    assert abs(rec['step.learningRate'] - p['learningRate'] * p['lrDecayRate']) < 1e-15
    report.append('trainIteration: max |dW - adam(oracle)| %.1e, runningLoss, lr ok' % worst_w)

    # ---- Model:retrieveBatch + utils.computeRanks on the UPDATED parameters (disc)
    if dec == 'disc':
        P1 = {k: v for k, v in W1.items()}
        ev = vo.forward_backward(enc, dec, P1, p, batch, None, only_forward=True)
        index(model, 'params').set('useGt', True)
        gt = to_py(first(inv(model, 'retrieveBatch', lua_batch())))
        index(model, 'params').set('useGt', False)
        allr = to_py(first(inv(model, 'retrieveBatch', lua_batch())))
        ref_scores = np.array(index(index(model, 'decoder'), 'output').a, dtype=np.float64)      # the reference decoder's own scores
        assert np.abs(ref_scores - ev['scores']).max() < 1e-12
        # utils.computeRanks on the reference's scores == the oracle's rule on the same scores, exactly; on the ORACLE's scores the ranks may
        # differ only where two candidates tie (duplicate options: equal to the last bit in one implementation, 1 ulp apart in the other)
        assert (np.asarray(allr).reshape(ref_scores.shape) == vo.compute_ranks(ref_scores)).all()
        assert (np.asarray(gt).reshape(-1) == vo.compute_ranks(ref_scores, batch['answer_ind'].reshape(-1) - 1)).all()
        from conftest import unexplained_rank_flips
        flipped, bad = unexplained_rank_flips(ev['scores'], ref_scores, tol=1e-9)
        assert not bad, bad[:5]
        rec['rank.scores'] = ref_scores
        rec['rank.gt'] = np.asarray(gt, dtype=np.int64).reshape(-1)
        rec['rank.all'] = np.asarray(allr, dtype=np.int64).reshape(ref_scores.shape)
        report.append('retrieveBatch / utils.computeRanks: gt + all ranks equal')
    # ---- Model:retrieveBatch, gen branch (model.lua:392-420): forwardConnect + decoder forward + utils.computeLhood per candidate, then
    #      utils.computeRanks, on the UPDATED parameters
    if dec == 'gen':
        P1 = {k: v for k, v in W1.items()}
        rb = {k: v for k, v in batch.items() if isinstance(v, np.ndarray)}
        dl.add_gen_options(rb, rb['ques_fwd'].shape[0])
        types = dict(BATCH_TYPES)
        lua_rb = lambda: to_lua(vm, rb, types)
        scores = vo.retrieve(enc, dec, P1, p, rb)                     # [N, O] summed token log-likelihoods
        index(model, 'params').set('useGt', True)
        gt = to_py(first(inv(model, 'retrieveBatch', lua_rb())))
        index(model, 'params').set('useGt', False)
        allr = np.asarray(to_py(first(inv(model, 'retrieveBatch', lua_rb())))).reshape(scores.shape)
        want = vo.compute_ranks(scores)
        flipped = int((allr != want).sum())
        if flipped:           # only where two candidates' likelihoods tie to the last bits (duplicate candidates)
            from conftest import unexplained_rank_flips
            order = np.argsort(allr, axis=1)                                  # the reference's order, as pseudo-scores
            pseudo = np.empty_like(scores)
            pseudo[np.arange(scores.shape[0])[:, None], order] = -np.arange(scores.shape[1], dtype=np.float64)[None, :]
            _, bad = unexplained_rank_flips(pseudo, scores, tol=1e-9)
            assert not bad, bad[:5]
        assert np.abs(np.asarray(gt).reshape(-1) - vo.compute_ranks(scores, rb['answer_ind'].reshape(-1) - 1)).sum() <= flipped
        rec.update({'rbatch.option_in': rb['option_in'], 'rbatch.option_out': rb['option_out'], 'rbatch.answer_ind': rb['answer_ind'],
                    'rank.scores': scores, 'rank.gt': np.asarray(gt, dtype=np.int64).reshape(-1), 'rank.all': allr.astype(np.int64)})
        report.append('retrieveBatch (gen: computeLhood x %d candidates) / computeRanks: equal (%d tie flips)' % (scores.shape[1], flipped))
    # ---- Model:generateAnswers, beam search (model.lua:432-573), one dialog, on the UPDATED parameters: token-exact against the oracle
    if dec == 'gen':
        START, END = p['vocabSize'] - 1, p['vocabSize']
        words = {i: 'w%d' % i for i in range(1, p['vocabSize'] + 1)}
        words[START], words[END] = '<START>', '<END>'
        gbatch = {k: v[:1] for k, v in batch.items() if isinstance(v, np.ndarray) and k in ('ques_fwd', 'hist', 'img_feat')}
        gbatch.update({k: batch[k][:1] for k in ('answer_in', 'answer_out')})

        class GenLoader(object):
            lua_type = 'table'

            def __init__(self):
                self.f = {'word2ind': to_lua(vm, {'<START>': START, '<END>': END}), 'ind2word': to_lua(vm, words),
                          'numThreads': to_lua(vm, {'val': 1}), 'unique_img_val': to_lua(vm, [4711])}

            def lua_index(self, k):
                if k == 'getIndexData':
                    return lambda *_a: to_lua(vm, gbatch, BATCH_TYPES)
                return self.f.get(k)
        beam_size, beam_len = 5, 12
        # model.lua:577 indexes finishBeams[1]: every round must finish a beam, as it does with a trained model.  A freshly initialised
        # head ranks the words almost identically at every step, so <END> is nudged into contention (a parameter like any other: the
        # oracle gets the same vector)
        W1 = named(flat_W)
        for bias in (0.04, 0.06, 0.08, 0.1, 0.13, 0.16, 0.2, 0.3):         # the smallest nudge with which every round finishes a beam
            trial = dict(W1)
            trial['vocab.b'] = W1['vocab.b'].copy()
            trial['vocab.b'][END - 1] += bias
            got = vo.generate_beam(enc, trial, p, gbatch, beam_size, beam_len, START, END)
            if all(END in toks for toks, _ in got):
                break
        else:
            raise AssertionError('no <END> bias finishes every round')
        for w, name in order:
            if name == 'vocab.b':
                w.a[END - 1] += bias
        W1 = named(flat_W)
        rec['beam.vocab_b'] = W1['vocab.b'].copy()
        ans = to_py(first(inv(model, 'generateAnswers', GenLoader(), 'val', to_lua(vm, {'beamSize': beam_size, 'beamLen': beam_len, 'maxThreads': 1}))))
        assert len(ans) == 1 and ans[0]['image_id'] == 4711 and len(ans[0]['dialog']) == 10
        want = vo.generate_beam(enc, W1, p, gbatch, beam_size, beam_len, START, END)
        texts = []
        for r, (toks, score) in enumerate(want):
            sent = ''
            for t in toks:                       # utils.idToWords (utils.lua:48-63)
                if t > 0:
                    sent += ' ' + words[int(t)]
                    if words[int(t)] == '<END>':
                        break
            assert ans[0]['dialog'][r]['answer'] == sent, (enc, r, ans[0]['dialog'][r]['answer'], sent)
            texts.append(sent)
        rec['beam.tokens'] = np.array([t for t, _ in want], dtype=np.int64)
        rec['beam.scores'] = np.array([sc for _, sc in want], dtype=np.float64)
        rec['beam.params'] = np.array([beam_size, beam_len, START, END], dtype=np.int64)
        report.append('generateAnswers (beam %d x %d): 10 rounds token-exact' % (beam_size, beam_len))
    # ---- the split-level loops: Model:evaluate, Model:retrieve (+ utils.processRanks), Model:predict (model.lua:109-246) over a
    #      3-dialog validation split served as two batches (2 + 1), on the parameters as they are now
    if (enc, dec) in SPLIT_PAIRS:
        flat_W.a[...] = flat_W.a.astype(np.float32).astype(np.float64)      # fp32-representable again (the Adam step was taken in fp64)
        Pn = named(flat_W)
        p2 = dict(p, decoder='gen')
        sb = []
        for nb in (2, 1):
            b = dl.getTrainBatch(p, batch_size=nb, full_length=False)
            if 'answer_out' not in b:                          # dataloader.lua:398-421: every batch carries the answers, whatever the decoder
                g = SyntheticDataloader(p2, seed=100 + nb).getTrainBatch(p2, batch_size=nb)
                b['answer_in'], b['answer_out'] = g['answer_in'], g['answer_out']
            if dec == 'gen':
                dl.add_gen_options(b, nb)
            sb.append({k: v for k, v in b.items() if isinstance(v, np.ndarray)})

        class SplitLoader(object):
            lua_type = 'table'

            def __init__(self):
                self.f = {'numThreads': to_lua(vm, {'val': 3}), 'unique_img_val': to_lua(vm, [11, 12, 13]),
                          'val_num_rounds': vm.torch.tensor(np.array([10, 9, 10]), 'Long')}

            def lua_index(self, k):
                if k == 'getTestBatch':
                    return lambda _self, start, *_a: (to_lua(vm, sb[0 if int(start) == 1 else 1], BATCH_TYPES), 3 if int(start) == 1 else 4)
                return self.f.get(k)
        D = SplitLoader()
        vm.captured.on, vm.captured.text = True, ''
        inv(model, 'evaluate', D, 'val')
        m_ev = re.search(r'val\tLoss: ([-0-9.]+)\t Perplexity: ([-0-9.a-z+]+)', vm.captured.text)
        vm.captured.text = ''
        index(model, 'params').set('useGt', True)
        recs = to_py(first(inv(model, 'retrieve', D, 'val')))
        txt = vm.captured.text
        vm.captured.text = ''
        index(model, 'params').set('useGt', False)
        pred = to_py(first(inv(model, 'predict', D, 'val')))
        vm.captured.on = False
        # oracle-side restatement of the three loops
        tot, ntok, gt_ranks, all_ranks = 0.0, 0, [], []
        for b in sb:
            tot += vo.forward_backward(enc, dec, Pn, p, b, None, only_forward=True)['loss']
            ntok += int((b['answer_out'] > 0).sum())
            sc = vo.retrieve(enc, dec, Pn, p, b)
            gt_ranks.append(vo.compute_ranks(sc, b['answer_ind'].reshape(-1) - 1).reshape(-1, 10))
            all_ranks.append(vo.compute_ranks(sc).reshape(-1, 10, sc.shape[1]))
        gt_ranks, all_ranks = np.concatenate(gt_ranks), np.concatenate(all_ranks)
        loss = tot / ntok
        assert m_ev and abs(float(m_ev.group(1)) - loss) < 1e-6 and abs(float(m_ev.group(2)) - np.exp(loss)) < 1e-5 * np.exp(loss), (vm.captured.text, loss)
        want = vo.process_ranks(gt_ranks)
        printed = {k: float(v) for k, v in re.findall(r'\t([a-zA-Z@0-9. ]+): ([-0-9.]+)', txt)}
        for k in ('r@1', 'r@5', 'r@10', 'medianR', 'meanR', 'meanRR'):
            assert abs(printed[k] - want[k]) < 1e-6, (k, printed, want)
        assert int(printed['No. questions']) == 30
        rounds = [10, 9, 10]
        exp_recs = [{'image_id': 11 + i, 'round_id': j + 1, 'ranks': float(gt_ranks[i, j])} for i in range(3) for j in range(rounds[i])]
        # utils.processRanks ends with torch.mean(ranks:cinv()) -- IN PLACE -- on `ranks:double():view(-1)`, which on the CPU path
        # (-gpuid -1: torch.Tensor is already a DoubleTensor, :double() returns it) is the very tensor Model:retrieve then turns into its
        # records: they hold 1 / rank there.  With -gpuid >= 0 the tensor is a CudaTensor, :double() copies, and the records hold the
        # ranks -- the behaviour this repo reproduces.  The run here is the CPU path, so:
        assert len(recs) == len(exp_recs) and all(a['image_id'] == b['image_id'] and a['round_id'] == b['round_id'] and
                                                  abs(1.0 / a['ranks'] - b['ranks']) < 1e-9 for a, b in zip(recs, exp_recs))
        exp_pred = [{'image_id': 11 + i, 'round_id': j + 1, 'ranks': [float(x) for x in all_ranks[i, j]]} for i in range(3) for j in range(rounds[i])]
        assert pred == exp_pred
        for i, b in enumerate(sb):
            rec.update({'split.batch%d.%s' % (i, k): v for k, v in b.items()})
        rec.update({'split.param.' + k: v.astype(np.float32) for k, v in Pn.items()})
        assert all((rec['split.param.' + k].astype(np.float64) == v).all() for k, v in Pn.items())
        rec['split.loss'], rec['split.ppl'] = np.float64(float(m_ev.group(1))), np.float64(float(m_ev.group(2)))
        rec['split.gt_ranks'], rec['split.all_ranks'] = gt_ranks.astype(np.int64), all_ranks.astype(np.int64)
        rec['split.metrics'] = np.array([printed[k] for k in ('r@1', 'r@5', 'r@10', 'medianR', 'meanR', 'meanRR')])
        report.append('evaluate / retrieve (+ processRanks) / predict over a 3-dialog split: loss %.6f, metrics and %d + %d records equal' % (
            loss, len(recs), len(pred)))
    # ---- two MORE Model:trainIteration calls on fresh batches: Adam's moments and bias correction at t = 2, 3, the learning-rate decay
    #      and the runningLoss moving average across iterations (model.lua:66-106, optim_updates.lua:62-91)
    if (enc, dec) in SPLIT_PAIRS:
        state = {}
        vo.train_iteration(enc, dec, P, p, batch, None, state, p['learningRate'])          # the oracle's optimiser state after the first call
        Pc = named(flat_W)                                                                   # (the parameters have been edited since: the state has not)
        rec.update({'multi.param.' + k: v.astype(np.float32) for k, v in Pc.items()})
        assert all((rec['multi.param.' + k].astype(np.float64) == v).all() for k, v in Pc.items())
        lr = float(index(index(model, 'optims'), 'learningRate'))
        rl = float(vm.globals.get('runningLoss'))
        rec['multi.start'] = np.array([lr, rl])
        inv(wrapper, 'evaluate')
        for it in range(2):
            b = dl.getTrainBatch(p)
            cur = {'b': b}

            class NextBatch(object):
                lua_type = 'table'

                def lua_index(self, k):
                    return (lambda *_a: to_lua(vm, {kk: v for kk, v in cur['b'].items() if isinstance(v, np.ndarray)}, BATCH_TYPES)) if k == 'getTrainBatch' else None
            inv(model, 'trainIteration', NextBatch())
            Pc, r = vo.train_iteration(enc, dec, Pc, p, b, None, state, lr)
            if lr > p['minLRate']:
                lr *= p['lrDecayRate']
            curl = r['loss'] / max(int((b['answer_out'] > 0).sum()), 1) if dec == 'gen' else r['loss']
            rl = 0.95 * rl + 0.05 * curl if rl > 0 else curl
            rec.update({'multi.batch%d.%s' % (it, k): v for k, v in b.items() if isinstance(v, np.ndarray)})
        Wn = named(flat_W)
        worst = max(float(np.abs(Wn[k] - Pc[k]).max()) for k in Pc)
        assert worst < 1e-12, (enc, dec, 'three iterations', worst, {k: float(np.abs(Wn[k] - Pc[k]).max()) for k in Pc})
        assert abs(float(vm.globals.get('runningLoss')) - rl) < 1e-9 * max(1, abs(rl))
        assert abs(float(index(index(model, 'optims'), 'learningRate')) - lr) < 1e-15
        rec.update({'multi.delta.' + k: (Wn[k] - rec['multi.param.' + k].astype(np.float64)).astype(np.float32) for k in Wn})
        rec['multi.end'] = np.array([lr, rl])
        report.append('trainIteration x 3 (Adam t = 2, 3; lr decay; runningLoss EMA): max |dW| vs oracle %.1e' % worst)
    return p, rec, names_in_flat_order, report


SPLIT_PAIRS = [('mn-att-ques-im-hist', 'disc'), ('lf-ques', 'gen'), ('hre-ques-im-hist', 'disc')]
VARIANTS = [('mn-att-ques-im-hist', 'disc', {'numAttentionLayers': 3}), ('lf-att-ques-im-hist', 'disc', {'numAttentionLayers': 3}),
            ('hrea-ques-im-hist', 'gen', {'numLayers': 3}), ('hre-ques-hist', 'gen', {'numLayers': 1}), ('lf-ques-im-hist', 'gen', {'numLayers': 1}),
            ('mn-ques-im-hist', 'gen', {'numLayers': 3})]


def variant_orders():
    """getParameters() order at other numLayers / numAttentionLayers (construction only, no step)"""
    return [{'encoder': enc, 'decoder': dec, 'params': extra, 'order': run_pair(enc, dec, extra=extra, order_only=True)[2]}
            for enc, dec, extra in VARIANTS]


def main():
    path = os.path.join(ROOT, 'tests', 'golden', 'reference_param_order.json')
    if '--variants-only' in sys.argv:
        d = json.load(open(path))
        d['variants'] = variant_orders()
        json.dump(d, open(path, 'w'), indent=1)
        return
    orders = {}
    for enc, dec in PAIRS:
        p, rec, order, report = run_pair(enc, dec)
        orders.setdefault(enc, {})[dec] = order
        print('%-22s + %-4s  %s' % (enc, dec, ' | '.join(report)), flush=True)
        print('    getParameters() order:', ' '.join(order), flush=True)
        np.savez_compressed(os.path.join(ROOT, 'tests', 'golden', 'ref__%s__%s.npz' % (enc, dec)), **rec)
    enc_orders = {}
    for enc, d in orders.items():
        o = d['disc']
        enc_orders[enc] = [n for n in o if not n.startswith('opt.')]
    json.dump({'_comment': "Order of the tensors in wrapper:getParameters() as produced by EXECUTING the reference's encoder / decoder files under "
                           "tests/luavm + tests/luavm/nn7.py (tests/golden/make_reference_goldens.py).  For the four nngraph encoders this is nngraph's "
                           "forward-node order (depth-first post-order from the output node) with shared storages at their first occurrence.  "
                           "DERIVED by running the reference's sources on a restated nngraph, not verified against a Torch7-written file.",
               'encoder': enc_orders, 'decoder': {'disc': ['opt.W', 'opt.b'], 'gen': [n for n in orders['lf-ques']['gen'] if n.startswith(('dec', 'vocab'))]},
               'variants': variant_orders()},
              open(path, 'w'), indent=1)


if __name__ == '__main__':
    main()
