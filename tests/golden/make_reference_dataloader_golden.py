"""EXECUTE the reference's own dataloader.lua + utils.lua (from /root/reference, in THIS container only) under tests/luavm and record what it
produces from a small random raw dataset with the reference's edge cases (zero-length question in the middle of a dialog, empty answers /
options, over-long captions, both history modes):

  * dataloader:initialize -> prepareDataset: ques_fwd (utils.rightAlign), processAnswers, processOptions, processHistory (dataloader.lua:10-318)
  * dataloader:getTestBatch('val') for the disc and the gen decoder, getIndexData, getTrainBatch with the drawn thread ids
    (dataloader.lua:321-475)

The hdf5 and cjson rocks are absent: `hdf5.open(...):read(name):all()` is served from the arrays below (tests/luavm stub), the JSON goes
through a real temporary file and a json-backed cjson stub.  At generation time every array is checked against oracle/dataloader_oracle.py
and against the product loader (visdial_amd.dataloader.Dataloader); tests/test_reference_dataloader.py keeps both on the stored outputs.
Output: tests/golden/ref_dataloader__<case>.npz (data only).   python tests/golden/make_reference_dataloader_golden.py [case ...]

Case `prepro` feeds dataloader.lua the file the reference's OWN data/prepro.py wrote (tests/golden/prepro/, made by make_prepro_golden.py
running the real prepro.py): reference writer -> reference reader, with only the hdf5 rock between them stubbed (arrays = the h5py
read-back of that file, expected.npz).  10 rounds, 100 options, truncated questions / answers / captions, UNK tokens, an empty answer."""
import json
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
REF = '/root/reference'

from luavm import LuaTable, new_vm, to_lua, to_py          # noqa: E402
from luavm.interp import call, index                       # noqa: E402
from test_dataloader_cpu import raw_dataset                # noqa: E402
from visdial_amd.dataloader import Dataloader              # noqa: E402
from visdial_amd.opts import default_params, derive        # noqa: E402

CASES = {'lf-ques-im-hist': dict(encoder='lf-ques-im-hist', concatHistory=False, att=False),
         'mn-att-concat': dict(encoder='mn-att-ques-im-hist', concatHistory=True, att=True),
         'lf-ques': dict(encoder='lf-ques', concatHistory=False, att=False),
         'prepro': dict(encoder='lf-ques-im-hist', concatHistory=False, att=False, prepro=True)}
PREPRO = os.path.join(ROOT, 'tests', 'golden', 'prepro')
first = lambda vals: vals[0] if vals else None


def two_splits(rng, att, R=4, O=5):
    info, raw, img = raw_dataset(rng, n=6, R=R, MQ=6, MA=5, V=30, O=O, nopt=40, F=8, att=att)
    info2, raw2, img2 = raw_dataset(rng, n=5, R=R, MQ=6, MA=5, V=30, O=O, nopt=40, F=8, att=att)
    raw.update({k.replace('_train', '_val'): v for k, v in raw2.items()})
    img.update({k.replace('_train', '_val'): v for k, v in img2.items()})
    info['unique_img_val'] = ['%012d' % (100 + i) for i in range(5)]
    return info, raw, img


class H5File(object):
    """hdf5.open(path, 'r'): `:read(name):all()` -> tensor of the array stored under that name (leading '/' ignored)"""
    lua_type = 'table'

    def __init__(self, vm, arrays):
        self.vm, self.arrays = vm, arrays

    def lua_index(self, k):
        if k == 'read':
            def read(_self, name, *_a):
                a = self.arrays[name.lstrip('/')]
                tn = 'Float' if a.dtype.kind == 'f' else 'Int'        # (the prepro files hold uint32 / float32)
                t = self.vm.torch.tensor(np.ascontiguousarray(a).astype(np.float32 if tn == 'Float' else np.int32), tn)
                box = LuaTable()
                box.set('all', lambda *_x: t)
                return box
            return read
        if k == 'close':
            return lambda *_a: None
        return None


def prepro_source():
    """what data/prepro.py itself wrote (make_prepro_golden.py): the params JSON as is, the datasets as h5py read them back"""
    z = np.load(os.path.join(PREPRO, 'expected.npz'))
    info = json.load(open(os.path.join(PREPRO, 'visdial_params.json')))
    raw = {k[5:]: z[k] for k in z.files if k.startswith('data.')}
    img = {k[4:]: z[k] for k in z.files if k.startswith('img.')}
    return info, raw, img


def run_case(name, cfg, seed=3):
    rng = np.random.RandomState(seed)
    info, raw, img = prepro_source() if cfg.get('prepro') else two_splits(rng, cfg['att'])
    opt = derive(default_params(encoder=cfg['encoder'], decoder='disc', batchSize=3, concatHistory=cfg['concatHistory']))
    tmp = tempfile.mkdtemp()
    jpath = os.path.join(tmp, 'params.json')
    json.dump(info, open(jpath, 'w'))                                  # ONE line: utils.readJSON reads a single line (utils.lua:67-68)
    vm = new_vm(search=[REF, os.path.join(ROOT, 'tests', 'lua_ref_stubs')], stdout=open(os.devnull, 'w'))
    cj = LuaTable()
    cj.set('decode', lambda text, *_: to_lua(vm, json.loads(text)))
    vm.preload.set('cjson', lambda *_: cj)
    h5 = LuaTable()
    files = {'ques.h5': raw, 'img.h5': img}
    h5.set('open', lambda path, *_a: H5File(vm, files[os.path.basename(path)]))
    vm.preload.set('hdf5', lambda *_: h5)
    vm.globals.set('hdf5', h5)
    dl = first(vm.dostring("return require 'dataloader'"))
    lopt = {k: v for k, v in opt.items() if isinstance(v, (int, float, str, bool))}
    lopt.update(inputJson=jpath, inputQues='ques.h5', inputImg='img.h5', gpuid=-1)
    call(index(dl, 'initialize'), [dl, to_lua(vm, lopt), to_lua(vm, ['train', 'val'])])
    rec = {'raw.' + k: v for k, v in raw.items()}
    rec.update({'img.' + k: v for k, v in img.items()})
    rec['info.json'] = np.array(json.dumps(info))
    rec['opt.json'] = np.array(json.dumps({k: v for k, v in opt.items() if isinstance(v, (int, float, str, bool))}))

    arr = lambda t: np.array(t.a)
    got = lambda key: arr(index(dl, key))
    # ---- prepareDataset outputs, both splits
    prod = Dataloader(seed=1).from_arrays(json.loads(json.dumps(info)), raw, img, opt, ['train', 'val'])
    fields = ['ques_fwd', 'ans_in', 'ans_out', 'ans_len', 'opt_in', 'opt_out', 'opt_len'] + (['hist', 'hist_len'] if opt['useHistory'] else [])
    for split in ('train', 'val'):
        for f in fields:
            ref = got('%s_%s' % (split, f))
            np.testing.assert_array_equal(ref, prod.data[split][f], err_msg='%s %s %s' % (name, split, f))
            rec['prep.%s.%s' % (split, f)] = ref.astype(np.int64)
    for k in ('vocabSize', 'maxQuesCount', 'maxQuesLen', 'maxAnsLen', 'numOptions', 'maxHistoryLen'):
        assert int(index(dl, k)) == int(getattr(prod, k)), (name, k, index(dl, k), getattr(prod, k))
    rec['stats'] = np.array([int(index(dl, k)) for k in ('vocabSize', 'maxQuesCount', 'maxQuesLen', 'maxAnsLen', 'numOptions', 'maxHistoryLen')])
    want_ids = [2001, 2002, 2003, 2004] if cfg.get('prepro') else [100, 101, 102, 103, 104]
    assert to_py(index(dl, 'unique_img_val')) == want_ids                            # tonumber(string.match(v, '000%d+'))
    rec['unique_img_val'] = np.array(want_ids)

    # ---- batches: getTestBatch (disc, gen), getIndexData, getTrainBatch with the drawn ids
    def batch_arrays(t):
        return {k: np.array(v.a) for k, v in to_py_shallow(t).items()}

    def to_py_shallow(t):
        out, k = {}, None
        while True:
            k, v = t.next(k)
            if k is None:
                return out
            out[k] = v
    params = dict(lopt, numTrainThreads=int(index(dl, 'numTrainThreads')))
    for dec in ('disc', 'gen'):
        p = dict(params, decoder=dec)
        start = 1
        for bi in range(2):                                   # 5 (prepro: 4) val threads, batch 3: a full and a short batch
            res = call(index(dl, 'getTestBatch'), [dl, start, to_lua(vm, p), 'val'])
            b, nxt = batch_arrays(res[0]), int(res[1])
            mine, mynxt = prod.getTestBatch(start, dict(opt, decoder=dec), 'val')
            assert nxt == mynxt
            assert set(b) == set(k for k, v in mine.items() if isinstance(v, np.ndarray)), (name, dec, sorted(b), sorted(mine))
            for k, v in b.items():
                np.testing.assert_array_equal(v, mine[k], err_msg='%s getTestBatch %s %s' % (name, dec, k))
                rec['test.%s.%d.%s' % (dec, bi, k)] = v.astype(np.float32 if v.dtype.kind == 'f' else np.int64)
            start = nxt
    # getTrainBatch: the thread ids come from torch's generator (inds:random(1, numTrainThreads)); pin them
    drawn = np.array([4, 1, 5] if cfg.get('prepro') else [4, 1, 6], dtype=np.int64)

    class FixedRng(object):
        def randint(self, lo, hi=None, size=None, **_k):
            return drawn[:int(np.prod(size))].reshape(size) if size is not None else int(drawn[0])

        def __getattr__(self, n):
            raise AttributeError(n)
    real_rng, vm.torch.rng = vm.torch.rng, FixedRng()
    for dec in ('disc', 'gen'):
        p = dict(params, decoder=dec)
        b = batch_arrays(first(call(index(dl, 'getTrainBatch'), [dl, to_lua(vm, p)])))
        mine = prod.getIndexBatch(drawn, dict(opt, decoder=dec), 'train') if hasattr(prod, 'getIndexBatch') else None
        if mine is None:
            mine = prod.getIndexData(drawn, dict(opt, decoder=dec), 'train')
            if dec == 'disc':
                o = prod.getIndexOption(drawn, dict(opt, decoder=dec), 'train')
                mine['options'] = o.reshape(o.shape[0] * o.shape[1], o.shape[2], -1)
                mine['answer_ind'] = mine['answer_ind'].reshape(-1)
        for k, v in b.items():
            np.testing.assert_array_equal(v, mine[k], err_msg='%s getTrainBatch %s %s' % (name, dec, k))
            rec['train.%s.%s' % (dec, k)] = v.astype(np.float32 if v.dtype.kind == 'f' else np.int64)
    vm.torch.rng = real_rng
    rec['train.inds'] = drawn
    return rec


def main():
    for name, cfg in CASES.items():
        if len(sys.argv) > 1 and name not in sys.argv[1:]:
            continue
        rec = run_case(name, cfg)
        np.savez_compressed(os.path.join(ROOT, 'tests', 'golden', 'ref_dataloader__%s.npz' % name), **rec)
        print('%-18s %d arrays: prepareDataset (train, val), getTestBatch disc / gen x 2, getTrainBatch disc / gen == product loader' % (name, len(rec)), flush=True)


if __name__ == '__main__':
    main()
