"""EXECUTE the reference's unedited train.lua END TO END (from /root/reference, in THIS container only) under tests/luavm:
opts.lua (torch.CmdLine) -> dataloader.lua:initialize on a small raw dataset -> Model(modelParams) (model.lua + the encoder / decoder files,
on tests/luavm/nn7.py) -> the training loop (trainIteration x numEpochs x numIterPerEpoch, dropout ON, checkpoints every saveIter epochs,
train.lua:83-121) -> torch.save of model_epoch_%d.t7 / model_final.t7.

What is pinned from outside: the thread ids `torch.LongTensor(size):random(1, numTrainThreads)` draws (dataloader.lua:323) and nothing else
-- the Dropout noise is whatever the run drew (recorded per call).  At generation time the oracle replays the run (the product loader's
batches for the same ids, the recorded masks, clamp + adam + lr decay + runningLoss) and must land on the SAME final flat vector
(wrapper:getParameters() order) to 1e-11; tests/test_reference_train.py keeps that, and runs the HIP hosts through the same iterations.
Output: tests/golden/ref_train__<encoder>__<decoder>.npz (data only).   python tests/golden/make_reference_train_golden.py"""
import json
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
REF = '/root/reference'

from luavm import LuaTable, new_vm, to_lua, to_py          # noqa: E402
from luavm import nn7                                      # noqa: E402
from luavm.interp import call, index                       # noqa: E402
from make_reference_dataloader_golden import H5File, two_splits      # noqa: E402
from make_reference_goldens import DROP_SITES, dropouts              # noqa: E402
from oracle import visdial_oracle as vo                    # noqa: E402
from visdial_amd import t7                                 # noqa: E402
from visdial_amd.dataloader import Dataloader              # noqa: E402
from visdial_amd.opts import default_params, derive        # noqa: E402

CASES = [('lf-ques-im-hist', 'gen'), ('mn-att-ques-im-hist', 'disc')]
IDS = [[4, 1, 6], [2, 2, 5], [3, 6, 1], [5, 4, 2]]          # thread ids of the four iterations (6 threads, batch 3)
first = lambda vals: vals[0] if vals else None
# train.lua:22 makes torch.FloatTensor the default type on its CPU path (-gpuid -1).  The run below keeps the VM's tensors in double
# (torch.setdefaulttensortype is a no-op here): the point is the ALGORITHM of the unedited script, checked against an fp64 replay
TOL = 1e-11


def run_case(enc, dec, seed=5):
    att = 'att' in enc
    rng = np.random.RandomState(seed)
    info, raw, img = two_splits(rng, att, R=10, O=100)          # (model.lua:281 and decoders/disc.lua:10 hard-code 10 rounds and 100 options)
    tmp = tempfile.mkdtemp()
    jpath = os.path.join(tmp, 'params.json')
    json.dump(info, open(jpath, 'w'))
    sizes = dict(imgFeatureSize=8, imgSpatialSize=3, imgEmbedSize=8, embedSize=24 if att else 12, rnnHiddenSize=32, commonEmbeddingSize=12)
    argv = ['-encoder', enc, '-decoder', dec, '-inputJson', jpath, '-inputQues', 'ques.h5', '-inputImg', 'img.h5', '-gpuid', '-1',
            '-batchSize', '3', '-numEpochs', '2', '-saveIter', '1', '-savePath', os.path.join(tmp, 'ck') + '/']
    for k, v in sizes.items():
        argv += ['-' + k, str(v)]

    vm = new_vm(search=[REF, os.path.join(ROOT, 'tests', 'lua_ref_stubs')], stdout=open(os.devnull, 'w'))
    NN = nn7.install(vm, seed=7)
    cj = LuaTable()
    cj.set('decode', lambda text, *_: to_lua(vm, json.loads(text)))
    cj.set('encode', lambda t, *_: json.dumps(to_py(t)))
    vm.preload.set('cjson', lambda *_: cj)
    h5 = LuaTable()
    files = {'ques.h5': raw, 'img.h5': img}
    h5.set('open', lambda path, *_a: H5File(vm, files[os.path.basename(path)]))
    vm.preload.set('hdf5', lambda *_: h5)
    vm.globals.set('hdf5', h5)
    paths = LuaTable()
    paths.set('mkdir', lambda *_a: True)
    vm.globals.set('paths', paths)
    vm.torch.module.set('manualSeed', lambda *_a: None)                 # (train.lua:13; the generators below are pinned instead)
    vm.torch.module.set('setdefaulttensortype', lambda *_a: None)       # (train.lua:22; see TOL)
    draws = {'i': 0}

    class FixedRng(object):
        def randint(self, lo, hi=None, size=None, **_k):
            ids = np.array(IDS[draws['i']], dtype=np.int64)
            draws['i'] += 1
            return ids.reshape(size) if size is not None else int(ids[0])
    vm.torch.rng = FixedRng()
    saved = []

    def save(path, obj, *_a):
        o = to_py_shallow(obj)
        saved.append((os.path.basename(path), np.array(o['modelW'].a, dtype=np.float64).reshape(-1),
                      float(index(o['optims'], 'learningRate')) if 'optims' in o else None, o['modelParams']))
    vm.torch.module.set('save', save)
    captured = {}

    def on_model(m):
        captured['model'] = m
        captured['W0'] = np.array(index(m, 'wrapperW').a, dtype=np.float64).reshape(-1)
    vm.globals.set('__on_model', on_model)
    per_iter = []
    vm.globals.set('__after_iter', lambda m: per_iter.append((np.array(index(m, 'wrapperW').a, dtype=np.float64).reshape(-1),
                                                              np.array(index(m, 'wrapperdW').a, dtype=np.float64).reshape(-1))))
    vm.dostring("require 'model'\nlocal orig = Model\nModel = function(...) local m = orig(...); __on_model(m); "
                "local ti = m.trainIteration; m.trainIteration = function(self, dl) ti(self, dl); __after_iter(self) end; return m end")
    vm.globals.set('arg', to_lua(vm, argv))
    vm.dofile('train.lua')

    # ---- what the run left behind
    assert [s[0] for s in saved] == ['model_epoch_1.t7', 'model_epoch_2.t7', 'model_final.t7'] and draws['i'] == 4
    model = captured['model']
    mp = {k: v for k, v in to_py_shallow(saved[-1][3]).items() if isinstance(v, (int, float, str, bool))}
    opt = derive(default_params(encoder=enc, decoder=dec, batchSize=3, **sizes))
    for k in ('useHistory', 'useIm', 'concatHistory', 'imgNorm', 'numLayers', 'dropout', 'learningRate', 'lrDecayRate', 'minLRate',
              'maxHistoryLen', 'numAttentionLayers', 'embedSize', 'rnnHiddenSize'):
        assert (mp.get(k) or 0) == (opt.get(k) or 0), (k, mp.get(k), opt.get(k))            # opts.lua's parse + derivations == opts.py
    prod = Dataloader(seed=1).from_arrays(json.loads(json.dumps(info)), raw, img, opt, ['train'])
    for k in ('vocabSize', 'maxQuesCount', 'maxQuesLen', 'maxAnsLen'):
        assert int(mp[k]) == int(getattr(prod, k))
        opt[k] = int(mp[k])
    assert int(mp['numIterPerEpoch']) == 2 and int(mp['numTrainThreads']) == 6
    opt['numOptions'] = int(prod.numOptions)

    # ---- the oracle replays it: flat vector -> named (the reference's getParameters() order), batches from the product loader
    spec = vo.param_spec(enc, dec, opt)
    P = {k: v.copy() for k, v in _split64(captured['W0'], spec, enc).items()}          # (t7.flat_to_named would round to fp32)
    drops = sorted(dropouts(index(model, 'wrapper')), key=lambda m: m.created)
    sites = DROP_SITES.get(enc, ['fuse'] if opt.get('dropout', 0.5) > 0 else [])
    assert len(drops) == len(sites)
    site_of = {id(d): s for d, s in zip(drops, sites)}
    log = NN.dropout_log
    assert len(log) == 4 * len(sites), (len(log), len(sites))
    state, lr, rl = {}, opt['learningRate'], 0.0
    rec = {}
    for it in range(4):
        ids = np.array(IDS[it], dtype=np.int64)
        b = prod.getIndexData(ids, opt, 'train')
        if dec == 'disc':
            o = prod.getIndexOption(ids, opt, 'train')
            b['options'] = o.reshape(o.shape[0] * o.shape[1], o.shape[2], -1)
            b['answer_ind'] = b['answer_ind'].reshape(-1)
        masks = {site_of[id(m)]: noise for m, noise in log[it * len(sites):(it + 1) * len(sites)]}
        P, r = vo.train_iteration(enc, dec, P, opt, b, {k: v.astype(np.float64) for k, v in masks.items()}, state, lr)
        if lr > opt['minLRate']:
            lr *= opt['lrDecayRate']
        cur = r['loss'] / max(int((b['answer_out'] > 0).sum()), 1) if dec == 'gen' else r['loss']
        rl = 0.95 * rl + 0.05 * cur if rl > 0 else cur
        rec.update({'it%d.mask.%s' % (it, k): v.astype(np.uint8) for k, v in masks.items()})
        if it in (1, 3):                       # checkpoints of epochs 1 and 2 (train.lua:90-102)
            ck = saved[0 if it == 1 else 1]
            flat64 = np.concatenate([P[n].reshape(-1) for n, _, _ in t7.reference_order(enc, spec)])
            assert np.abs(flat64 - ck[1]).max() < TOL, (enc, dec, it, float(np.abs(flat64 - ck[1]).max()))
            assert abs(ck[2] - lr) < 1e-15
    final = np.concatenate([P[n].reshape(-1) for n, _, _ in t7.reference_order(enc, spec)])
    worst = float(np.abs(final - saved[1][1]).max())                       # the epoch-2 checkpoint IS the final state, in double
    assert worst < TOL, (enc, dec, 'final', worst)
    f32 = final.astype(np.float32).astype(np.float64)                      # train.lua:120 saves model.wrapperW:float()
    assert (f32 != saved[-1][1]).mean() < 1e-3 and np.abs(f32 - saved[-1][1]).max() < 2e-7, (enc, dec, 'model_final.t7')
    assert abs(float(vm.globals.get('runningLoss')) - rl) < 1e-9 * max(1.0, abs(rl))
    rec.update({'raw.' + k: v for k, v in raw.items() if k.endswith('_train')})
    rec.update({'img.' + k: v for k, v in img.items() if k.endswith('_train')})
    rec['info.json'] = np.array(json.dumps({k: v for k, v in info.items() if not k.endswith('_val')}))
    rec['opt.json'] = np.array(json.dumps({k: v for k, v in opt.items() if isinstance(v, (int, float, str, bool))}))
    rec['ids'] = np.array(IDS, dtype=np.int64)
    rec['W0'] = captured['W0']                                  # (fp64 initialisation: stored exactly)
    rec['final.delta'] = (saved[1][1] - captured['W0']).astype(np.float32)
    rec['epoch1.delta'] = (saved[0][1] - captured['W0']).astype(np.float32)
    rec['end'] = np.array([lr, rl])

    # ---- evaluate.lua, unedited, on the epoch-2 checkpoint and the val split: -useGt true (Model:retrieve) and false (Model:predict),
    #      -saveRanks true (utils.writeJSON).  torch.load hands back what torch.save was given.
    ck = saved[1]
    loaded = LuaTable()
    loaded.set('modelW', vm.torch.tensor(ck[1].copy(), 'Double'))
    loaded.set('modelParams', ck[3])
    lo = LuaTable()
    lo.set('learningRate', ck[2])
    loaded.set('optims', lo)
    vm.torch.module.set('load', lambda path, *_a: loaded)
    Pf = _split64(ck[1], spec, enc)
    val = Dataloader(seed=1).from_arrays(json.loads(json.dumps(info)), raw, img, opt, ['val'])
    results = {}
    for use_gt in (True, False):
        out_json = os.path.join(tmp, 'ranks_%d.json' % use_gt)
        vm.globals.set('arg', to_lua(vm, ['-inputJson', jpath, '-inputQues', 'ques.h5', '-inputImg', 'img.h5', '-gpuid', '-1', '-loadPath',
                                          os.path.join(tmp, 'ck', 'model_epoch_2.t7'), '-split', 'val', '-batchSize', '2',
                                          '-useGt', 'true' if use_gt else 'false', '-saveRanks', 'true', '-saveRankPath', out_json]))
        vm.dofile('evaluate.lua')
        results[use_gt] = json.load(open(out_json))
    # the oracle's version of the same: batches of 2 + 2 + 1 dialogs from the product loader
    p_eval = dict(opt, batchSize=2)
    start, gt_r, all_r = 1, [], []
    while start <= 5:
        b, start = val.getTestBatch(start, p_eval, 'val')
        sc = vo.retrieve(enc, dec, Pf, opt, b)
        gt_r.append(vo.compute_ranks(sc, np.asarray(b['answer_ind']).reshape(-1) - 1).reshape(-1, 10))
        all_r.append(vo.compute_ranks(sc).reshape(-1, 10, sc.shape[1]))
    gt_r, all_r = np.concatenate(gt_r), np.concatenate(all_r)
    ids = [100 + i for i in range(5)]
    nr = raw['num_rounds_val'].astype(int)
    want = [(ids[i], j + 1) for i in range(5) for j in range(nr[i])]
    assert [(r['image_id'], r['round_id']) for r in results[True]] == want and [(r['image_id'], r['round_id']) for r in results[False]] == want
    # (-useGt true goes through utils.processRanks, which inverts the tensor in place on this CPU path: the records hold 1 / rank)
    assert all(abs(1.0 / r['ranks'] - gt_r[i, j]) < 1e-9 for r, (i, j) in zip(results[True], [(i, j) for i in range(5) for j in range(nr[i])]))
    assert all(r['ranks'] == [float(x) for x in all_r[i, j]] for r, (i, j) in zip(results[False], [(i, j) for i in range(5) for j in range(nr[i])]))
    rec.update({'raw.' + k: v for k, v in raw.items() if k.endswith('_val')})
    rec.update({'img.' + k: v for k, v in img.items() if k.endswith('_val')})
    rec['info.json'] = np.array(json.dumps(info))
    rec['eval.gt_ranks'], rec['eval.all_ranks'] = gt_r.astype(np.int64), all_r.astype(np.int64)

    # ---- generate.lua, unedited (gen decoder): beam search over the first two val dialogs, results.json.  NOTE generate.lua:57-70 sets
    #      useHistory / useIm but NOT concatHistory: the dataloader builds per-round history there even for the lf-* encoders.
    if dec == 'gen':
        V = int(opt['vocabSize'])
        START, END = V - 1, V
        gen_opt = dict(opt, concatHistory=False, maxHistoryLen=60)
        gval = Dataloader(seed=1).from_arrays(json.loads(json.dumps(info)), raw, img, gen_opt, ['val'])
        gb = [gval.getIndexData(np.array([c]), gen_opt, 'val') for c in (1, 2)]
        bs, bl = 5, 10
        for bias in (0.0, 0.05, 0.1, 0.2, 0.3, 0.5, 0.8):          # (model.lua:577 needs a finished beam in every round: see make_reference_goldens.py)
            Pg = {k: v.copy() for k, v in Pf.items()}
            Pg['vocab.b'][END - 1] += bias
            beams = [vo.generate_beam(enc, Pg, gen_opt, b, bs, bl, START, END) for b in gb]
            if all(END in toks for conv in beams for toks, _ in conv):
                break
        else:
            raise AssertionError('no <END> bias finishes every round')
        flatg = np.concatenate([Pg[n].reshape(-1) for n, _, _ in t7.reference_order(enc, spec)])
        loaded.set('modelW', vm.torch.tensor(flatg.copy(), 'Double'))
        res_dir = os.path.join(tmp, 'results')
        os.makedirs(res_dir, exist_ok=True)
        vm.globals.set('arg', to_lua(vm, ['-inputJson', jpath, '-inputQues', 'ques.h5', '-inputImg', 'img.h5', '-gpuid', '-1', '-loadPath',
                                          os.path.join(tmp, 'ck', 'model_epoch_2.t7'), '-resultPath', res_dir, '-beamSize', str(bs),
                                          '-beamLen', str(bl), '-maxThreads', '2']))
        vm.dofile('generate.lua')
        out = json.load(open(os.path.join(res_dir, 'results.json')))
        ind2word = {int(i): w for w, i in info['word2ind'].items()}
        ind2word[START], ind2word[END] = '<START>', '<END>'

        def words(vec):
            s_ = ''
            for t in vec:
                if t > 0:
                    s_ += ' ' + ind2word[int(t)]
                    if ind2word[int(t)] == '<END>':
                        break
            return s_
        assert [d['image_id'] for d in out['data']] == [100, 101]
        for c in range(2):
            assert [r['answer'] for r in out['data'][c]['dialog']] == [words(toks) for toks, _ in beams[c]], (c, out['data'][c]['dialog'])
            assert [r['question'] for r in out['data'][c]['dialog']] == [words(q) for q in gb[c]['ques_fwd'][0]]
        rec['gen.vocab_b'] = Pg['vocab.b'].copy()
        rec['gen.tokens'] = np.array([[toks for toks, _ in conv] for conv in beams], dtype=np.int64)
        rec['gen.params'] = np.array([bs, bl, START, END], dtype=np.int64)
    return rec, worst


def _split64(flat, spec, enc):
    out, o = {}, 0
    for n, shape, _ in t7.reference_order(enc, spec):
        k = int(np.prod(shape))
        out[n] = flat[o:o + k].reshape(shape)
        o += k
    return out


def to_py_shallow(t):
    out, k = {}, None
    while True:
        k, v = t.next(k)
        if k is None:
            return out
        out[k] = v


def main():
    for enc, dec in CASES:
        rec, worst = run_case(enc, dec)
        np.savez_compressed(os.path.join(ROOT, 'tests', 'golden', 'ref_train__%s__%s.npz' % (enc, dec)), **rec)
        print('%-22s + %-4s  train.lua end to end: opts.lua, dataloader.lua, Model, 4 iterations with dropout, 3 checkpoints; final flat '
              'vector vs the oracle replay: max |dW| %.1e; evaluate.lua on the checkpoint (-useGt true / false, -saveRanks): ranks equal%s'
              % (enc, dec, worst, '; generate.lua (beam search, results.json): answers token-exact' if dec == 'gen' else ''), flush=True)


if __name__ == '__main__':
    main()
