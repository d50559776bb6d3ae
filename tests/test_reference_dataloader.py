"""Fixtures made by EXECUTING the reference's own dataloader.lua + utils.lua under tests/luavm (tests/golden/make_reference_dataloader_golden.py;
the hdf5 / cjson rocks replaced by array- and json-backed stubs): what `dataloader:initialize` / `prepareDataset` build from a small raw
dataset (right-aligned questions, <START>/<END>-wrapped answers and options, caption + QA history in both modes) and the batches
`getTestBatch` (disc, gen; a full and a short batch), `getTrainBatch` (pinned thread ids) hand to the model -- SURVEY.md 8 rows a3 / f1.
The product loader (visdial_amd.dataloader.Dataloader) and the loop-by-loop oracle (oracle/dataloader_oracle.py) must reproduce every array
bit for bit.  Nothing here reads /root/reference."""
import glob
import json
import os

import numpy as np
import pytest

from conftest import ROOT
from oracle import dataloader_oracle as do
from visdial_amd.dataloader import Dataloader
from visdial_amd.opts import derive

FILES = sorted(glob.glob(os.path.join(ROOT, 'tests', 'golden', 'ref_dataloader__*.npz')))
IDS = [os.path.basename(f)[len('ref_dataloader__'):-4] for f in FILES]


def load(path):
    z = np.load(path)
    get = lambda pre: {k[len(pre):]: z[k] for k in z.files if k.startswith(pre)}
    return z, get, json.loads(str(z['info.json'])), derive(json.loads(str(z['opt.json']))), get('raw.'), get('img.')


def test_fixture_set():
    assert sorted(IDS) == ['lf-ques', 'lf-ques-im-hist', 'mn-att-concat', 'prepro']


@pytest.mark.parametrize("path", FILES, ids=IDS)
def test_product_loader_reproduces_the_executed_reference(path):
    z, get, info, opt, raw, img = load(path)
    dl = Dataloader(seed=1).from_arrays(info, raw, img, opt, ['train', 'val'])
    stats = [int(getattr(dl, k)) for k in ('vocabSize', 'maxQuesCount', 'maxQuesLen', 'maxAnsLen', 'numOptions', 'maxHistoryLen')]
    assert stats == z['stats'].tolist()
    assert list(dl.unique_img_val) == (z['unique_img_val'].tolist() if 'unique_img_val' in z.files else [100, 101, 102, 103, 104])
    n = 0
    for split in ('train', 'val'):
        for f, want in get('prep.%s.' % split).items():
            np.testing.assert_array_equal(dl.data[split][f], want, err_msg='%s %s' % (split, f))
            n += 1
    assert n >= 14
    for dec in ('disc', 'gen'):
        start = 1
        for bi in range(2):
            b, start = dl.getTestBatch(start, dict(opt, decoder=dec), 'val')
            want = get('test.%s.%d.' % (dec, bi))
            assert set(want) == {k for k, v in b.items() if isinstance(v, np.ndarray)}
            for k, v in want.items():
                np.testing.assert_array_equal(b[k], v, err_msg='getTestBatch %s %d %s' % (dec, bi, k))
        assert start == raw['ques_val'].shape[0] + 1                   # numValThreads + 1
        inds = z['train.inds']
        b = dl.getIndexData(inds, dict(opt, decoder=dec), 'train')
        if dec == 'disc':                                              # dataloader.lua:330-337
            o = dl.getIndexOption(inds, dict(opt, decoder=dec), 'train')
            b['options'] = o.reshape(o.shape[0] * o.shape[1], o.shape[2], -1)
            b['answer_ind'] = b['answer_ind'].reshape(-1)
        for k, v in get('train.%s.' % dec).items():
            np.testing.assert_array_equal(b[k], v, err_msg='getTrainBatch %s %s' % (dec, k))


@pytest.mark.parametrize("path", FILES, ids=IDS)
def test_oracle_reproduces_the_executed_reference(path):
    z, get, info, opt, raw, img = load(path)
    V = len(info['word2ind'])
    START, END = V + 1, V + 2
    for split in ('train', 'val'):
        L = lambda k: raw['%s_%s' % (k, split)].astype(np.int64)
        want = get('prep.%s.' % split)
        np.testing.assert_array_equal(do.right_align(L('ques'), L('ques_length')), want['ques_fwd'])
        din, dout, alen = do.process_answers(L('ans'), L('ans_length'), START, END)
        np.testing.assert_array_equal(din, want['ans_in']); np.testing.assert_array_equal(dout, want['ans_out'])
        np.testing.assert_array_equal(alen, want['ans_len'])
        oin, oout, olen = do.process_options(L('opt_list'), L('opt_length'), L('ans').shape[2], START, END)
        np.testing.assert_array_equal(oin, want['opt_in']); np.testing.assert_array_equal(oout, want['opt_out'])
        np.testing.assert_array_equal(olen, want['opt_len'])
        if opt['useHistory']:
            h, hl, _ = do.process_history(L('cap'), L('cap_length'), L('ques'), L('ques_length'), L('ans'), L('ans_length'),
                                          opt['concatHistory'], END)
            np.testing.assert_array_equal(h, want['hist']); np.testing.assert_array_equal(hl, want['hist_len'])
    if 'prepro' in path:
        return
    # the quirks the fixture is built to contain are live: rightAlign's `break` (a zero-length question hides the later ones of the dialog)
    ql = raw['ques_length_train'].astype(np.int64)
    assert (ql[1, 1] == 0) and (ql[1, 2:] > 0).any() and (get('prep.train.')['ques_fwd'][1, 1:] == 0).all()
