"""The input contract pinned by the reference's own code: tests/golden/prepro/ holds what the REAL
data/prepro.py wrote (tests/golden/make_prepro_golden.py runs the file itself as __main__ in the build container) and
tests/golden/ref_dataloader__prepro.npz what the reference's dataloader.lua built from it (executed under tests/luavm).  Here:
  * visdial_amd.h5lite (the product's HDF5 reader, ctypes over libhdf5) reads prepro.py's file bit for bit (vs the h5py read-back);
  * the product `Dataloader`, opened on the FILES the way train.py opens them, reproduces dataloader.lua's arrays and batches;
  * the properties of prepro.py's output the loaders rely on hold in the fixture (1-based options / ans_index, UNK, truncation, the test
    split's missing answers and options);
  * (-m gpu) train.py + evaluate.py run on the files end to end.
Nothing here reads /root/reference."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT
from visdial_amd import h5lite
from visdial_amd.dataloader import Dataloader
from visdial_amd.opts import default_params, derive

PRE = os.path.join(ROOT, 'tests', 'golden', 'prepro')
QUES, IMG, INFO = (os.path.join(PRE, n) for n in ('visdial_data.h5', 'data_img.h5', 'visdial_params.json'))
needs_hdf5 = pytest.mark.skipif(not h5lite.available(), reason="libhdf5 not loadable on this machine")


def expected(prefix):
    z = np.load(os.path.join(PRE, 'expected.npz'))
    return {k[len(prefix):]: z[k] for k in z.files if k.startswith(prefix)}


@needs_hdf5
def test_h5lite_reads_the_file_prepro_py_wrote():
    for path, prefix in ((QUES, 'data.'), (IMG, 'img.')):
        want = expected(prefix)
        with h5lite.File(path) as f:
            assert sorted(f.keys()) == sorted(want)
            for k, v in want.items():
                got = f[k][()] if hasattr(f[k], '__getitem__') else f[k]
                got = np.asarray(got)
                assert got.shape == v.shape and got.dtype == v.dtype, (k, got.dtype, v.dtype)
                np.testing.assert_array_equal(got, v, err_msg=k)
    assert len(expected('data.')) == 35            # 12 per train / val, 11 for test (no ans_index): prepro.py:263-277


def test_properties_of_the_reference_output():
    d = expected('data.')
    info = json.load(open(INFO))
    V = len(info['word2ind'])
    assert info['word2ind']['UNK'] == V and min(info['word2ind'].values()) == 1          # prepro.py:236-240: 1-based, UNK appended last
    assert {int(k): v for k, v in info['ind2word'].items()} == {v: k for k, v in info['word2ind'].items()}
    for s in ('train', 'val', 'test'):
        assert d['ques_' + s].shape[1:] == (10, 8) and d['ans_' + s].shape[1:] == (10, 6) and d['cap_' + s].shape[1] == 16
        assert d['ques_' + s].dtype == np.uint32 and d['ques_' + s].max() <= V
        assert d['ques_length_' + s].max() == 8 and d['cap_length_' + s].max() == 16     # truncated at -max_*_len (prepro.py:116,134,139)
        assert d['opt_' + s].min() >= 1 and d['opt_' + s].max() <= d['opt_list_' + s].shape[0]      # 1-based (prepro.py:149-150)
        assert (d['opt_length_' + s] == 0).any()                                         # the empty answer
        ql, q = d['ques_length_' + s], d['ques_' + s]
        assert ((q != 0).sum(-1) == ql).all()                                            # left-aligned, zero-padded
    assert (d['ques_train'] == V).any()                                                  # rare words became UNK
    assert d['ans_index_train'].min() >= 1 and d['ans_index_train'].max() <= 100 and 'ans_index_test' not in d
    # the test split: fewer rounds, the last round unanswered, options only for it (prepro.py:57-65,158-159)
    nr = d['num_rounds_test'].astype(int)
    assert nr.min() >= 1 and nr.max() <= 10 and (nr < 10).any()
    for i, r in enumerate(nr):
        assert d['ans_length_test'][i, r - 1] == 0 and (d['ques_length_test'][i, r:] == 0).all()
        assert (d['opt_test'][i, :r - 1] == 1).all() and (d['opt_test'][i, r - 1] > 1).any()
    gt = np.take_along_axis(d['opt_train'].astype(int), d['ans_index_train'].astype(int)[..., None] - 1, 2)[..., 0]
    assert (d['opt_list_train'][gt - 1] == d['ans_train']).all()                         # the ground-truth option IS the round's answer


@needs_hdf5
def test_product_loader_on_the_files_equals_the_executed_dataloader_lua():
    z = np.load(os.path.join(ROOT, 'tests', 'golden', 'ref_dataloader__prepro.npz'))
    opt = derive(json.loads(str(z['opt.json'])))
    opt.update(inputQues=QUES, inputImg=IMG, inputJson=INFO)
    dl = Dataloader(seed=1)
    dl.initialize(opt, ['train', 'val'])
    assert [int(getattr(dl, k)) for k in ('vocabSize', 'maxQuesCount', 'maxQuesLen', 'maxAnsLen', 'numOptions', 'maxHistoryLen')] == z['stats'].tolist()
    assert list(dl.unique_img_val) == z['unique_img_val'].tolist() == [2001, 2002, 2003, 2004]
    n = 0
    for k in z.files:
        if k.startswith('prep.'):
            _, split, f = k.split('.')
            np.testing.assert_array_equal(dl.data[split][f], z[k], err_msg=k)
            n += 1
    assert n == 18
    for dec in ('disc', 'gen'):
        start = 1
        for bi in range(2):
            b, start = dl.getTestBatch(start, dict(opt, decoder=dec), 'val')
            for k in [k for k in z.files if k.startswith('test.%s.%d.' % (dec, bi))]:
                np.testing.assert_array_equal(b[k.split('.', 3)[3]], z[k], err_msg=k)
        assert start == 5
        b = dl.getIndexData(z['train.inds'], dict(opt, decoder=dec), 'train')
        if dec == 'disc':
            o = dl.getIndexOption(z['train.inds'], dict(opt, decoder=dec), 'train')
            b['options'] = o.reshape(o.shape[0] * o.shape[1], o.shape[2], -1)
            b['answer_ind'] = b['answer_ind'].reshape(-1)
        for k in [k for k in z.files if k.startswith('train.%s.' % dec)]:
            np.testing.assert_array_equal(b[k.split('.', 2)[2]], z[k], err_msg=k)
    # the third split prepro.py writes also loads (test: no ans_index, ragged num_rounds)
    dl3 = Dataloader(seed=1)
    dl3.initialize(opt, ['test'])
    assert dl3.data['test']['ques_fwd'].shape == (3, 10, 8)


@pytest.mark.gpu
@needs_hdf5
def test_train_and_evaluate_on_the_files_prepro_py_wrote(tmp_path):
    """train.py / evaluate.py on the reference-written dataset (10 rounds, 100 options per round): training lowers the loss, evaluation
    ranks every round of the val split"""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    data = ['-inputQues', QUES, '-inputImg', IMG, '-inputJson', INFO]
    save = str(tmp_path / "ckpt") + "/"
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'train.py'), '-encoder', 'lf-ques-im-hist', '-decoder', 'disc',
                        '-imgFeatureSize', '16', '-rnnHiddenSize', '32', '-embedSize', '16', '-batchSize', '2', '-savePath', save,
                        '-numEpochs', '100', '-saveIter', '1000', '--maxIters', '300', '-saveFormat', 'pt'] + data,
                       capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert 'using synthetic' not in r.stdout
    loss = [float(l.split('[Loss:')[1].split(']')[0]) for l in r.stdout.splitlines() if '[Loss:' in l]
    assert len(loss) >= 2 and loss[-1] < loss[0]
    ranks = str(tmp_path / 'ranks.json')
    e = subprocess.run([sys.executable, os.path.join(ROOT, 'evaluate.py'), '-loadPath', save + 'model_final.pt', '-batchSize', '3',
                        '-split', 'val', '-saveRanks', '1', '-saveRankPath', ranks] + data,
                       capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert e.returncode == 0, e.stdout[-2000:] + e.stderr[-2000:]
    assert 'r@1:' in e.stdout and 'meanRR:' in e.stdout
    rec = json.load(open(ranks))
    assert len(rec) == 40 and rec[0]['image_id'] == 2001 and all(1 <= x['ranks'] <= 100 for x in rec)
