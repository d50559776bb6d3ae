"""Real-data loader (visdial_amd.dataloader.Dataloader, counterpart of dataloader.lua) against the
loop-by-loop oracle restatement, on a random raw dataset with the reference's edge cases: zero-length
question in the middle of a dialog (rightAlign's `break`), empty answers / options, over-long captions,
concatenated history.  Integer outputs must be bit-exact."""
import json

import numpy as np
import pytest

from oracle import dataloader_oracle as do
from visdial_amd.dataloader import Dataloader, rightAlign
from visdial_amd.opts import derive, default_params


def raw_dataset(rng, n=7, R=4, MQ=6, MA=5, V=30, O=5, nopt=40, F=8, att=False):
    def seqs(shape, maxlen, allow_zero):
        L = rng.randint(0 if allow_zero else 1, maxlen + 1, size=shape[:-1])
        a = np.zeros(shape, np.uint32)
        for idx in np.ndindex(*shape[:-1]):
            a[idx][:L[idx]] = rng.randint(1, V + 1, size=L[idx])
        return a, L.astype(np.uint32)
    ques, ql = seqs((n, R, MQ), MQ, True)
    ques[1, 1] = 0; ql[1, 1] = 0                                   # forces the rightAlign break on image 1
    ans, al = seqs((n, R, MA), MA, True)
    cap, cl = seqs((n, 20), 20, False)                             # longer than MQ+MA = 11
    optl, ol = seqs((nopt, MA), MA, True)
    d = {'ques_train': ques, 'ques_length_train': ql, 'ans_train': ans, 'ans_length_train': al,
         'cap_train': cap, 'cap_length_train': cl, 'opt_list_train': optl, 'opt_length_train': ol,
         'opt_train': rng.randint(1, nopt + 1, size=(n, R, O)).astype(np.uint32),
         'ans_index_train': rng.randint(1, O + 1, size=(n, R)).astype(np.uint32),
         'img_pos_train': rng.permutation(n).astype(np.uint32), 'num_rounds_train': np.full(n, R, np.uint32)}
    img = {'images_train': (np.abs(rng.randn(n, F, 3, 3)) if att else rng.randn(n, F)).astype(np.float32)}
    info = {'word2ind': {'w%d' % i: i for i in range(1, V + 1)}, 'unique_img_train': ['%012d' % i for i in range(n)]}
    return info, d, img


@pytest.mark.parametrize("enc", ['lf-ques-im-hist', 'hre-ques-im-hist', 'mn-att-ques-im-hist', 'lf-ques'])
def test_preprocessing_and_batches_match_oracle(enc):
    rng = np.random.RandomState(3)
    att = 'att' in enc
    info, raw, img = raw_dataset(rng, att=att)
    opt = derive(default_params(encoder=enc, decoder='disc', batchSize=3))
    dl = Dataloader(seed=5).from_arrays(info, raw, img, opt, ['train'])
    d = dl.data['train']
    V = 30
    START, END = V + 1, V + 2
    assert dl.vocabSize == V + 2 and dl.word2ind['<START>'] == START and dl.word2ind['<END>'] == END
    L = lambda k: raw[k].astype(np.int64)
    np.testing.assert_array_equal(d['ques_fwd'], do.right_align(L('ques_train'), L('ques_length_train')))
    assert (d['ques_fwd'][1, 1:] == 0).all() and (L('ques_length_train')[1, 2:] > 0).any()      # the break quirk is live
    din, dout, alen = do.process_answers(L('ans_train'), L('ans_length_train'), START, END)
    np.testing.assert_array_equal(d['ans_in'], din); np.testing.assert_array_equal(d['ans_out'], dout)
    np.testing.assert_array_equal(d['ans_len'], alen)
    oin, oout, olen = do.process_options(L('opt_list_train'), L('opt_length_train'), 5, START, END)
    np.testing.assert_array_equal(d['opt_in'], oin); np.testing.assert_array_equal(d['opt_out'], oout)
    np.testing.assert_array_equal(d['opt_len'], olen)
    if opt['useHistory']:
        h, hl, W = do.process_history(L('cap_train'), L('cap_length_train'), L('ques_train'), L('ques_length_train'),
                                      L('ans_train'), L('ans_length_train'), opt['concatHistory'], END)
        np.testing.assert_array_equal(d['hist'], h); np.testing.assert_array_equal(d['hist_len'], hl)
        assert dl.maxHistoryLen == (W if opt['concatHistory'] else 60)
    if opt['useIm']:
        f = img['images_train']
        if att:
            np.testing.assert_array_equal(d['img_fv'], f.transpose(0, 2, 3, 1))                 # NCHW -> NHWC, no norm
        else:
            np.testing.assert_allclose(np.linalg.norm(d['img_fv'], axis=1), 1.0, rtol=1e-5)
    # batch assembly for explicit (1-based) thread ids
    inds = np.array([2, 7, 2])
    ref = do.index_data(dict(d, ans_len1=d['ans_len']), inds - 1, opt['useHistory'], dl.maxHistoryLen)
    got = dl.getIndexData(inds, opt, 'train')
    for k in ref:
        np.testing.assert_array_equal(got[k], ref[k], err_msg=k)
    if opt['useIm']:
        np.testing.assert_array_equal(got['img_feat'], d['img_fv'][L('img_pos_train')[inds - 1]])
    o = dl.getIndexOption(inds, opt, 'train')
    assert o.shape == (3, 4, 5, 5)
    np.testing.assert_array_equal(o[1, 2, 3], L('opt_list_train')[L('opt_train')[6, 2, 3] - 1])
    b = dl.getTrainBatch(opt)
    assert b['options'].shape[:2] == (12, 5) and b['answer_ind'].shape == (12,) and b['ques_fwd'].shape[:2] == (3, 4)
    tb, nxt = dl.getTestBatch(6, dict(opt, decoder='gen'), 'train')
    assert nxt == 8 and tb['option_in'].shape[:3] == (2, 4, 5) and (tb['option_in'][..., 0] == START).all()
    assert tb['option_in'].shape == tb['option_out'].shape and 'num_rounds' in tb


def test_right_align_2d_and_product_vs_oracle_random():
    rng = np.random.RandomState(0)
    for _ in range(20):
        n, R, M = rng.randint(1, 6), rng.randint(1, 5), rng.randint(1, 9)
        L = rng.randint(0, M + 1, size=(n, R))
        s = rng.randint(1, 50, size=(n, R, M)) * (np.arange(M)[None, None, :] < L[:, :, None])
        np.testing.assert_array_equal(rightAlign(s, L), do.right_align(s, L))
        np.testing.assert_array_equal(rightAlign(s[:, 0], L[:, 0]), do.right_align(s[:, 0], L[:, 0]))


def test_initialize_from_files(tmp_path):
    rng = np.random.RandomState(1)
    info, raw, img = raw_dataset(rng)
    (tmp_path / 'p.json').write_text(json.dumps(info))
    np.savez(str(tmp_path / 'q.npz'), **raw)
    np.savez(str(tmp_path / 'i.npz'), **img)
    opt = derive(default_params(encoder='lf-ques-im-hist', decoder='gen', batchSize=2, inputJson=str(tmp_path / 'p.json'),
                                inputQues=str(tmp_path / 'q.npz'), inputImg=str(tmp_path / 'i.npz')))
    dl = Dataloader().initialize(opt, ['train'])
    assert dl.numTrainThreads == 7 and dl.maxQuesCount == 4 and dl.maxQuesLen == 6 and dl.maxAnsLen == 5
    b = dl.getTrainBatch(opt)
    assert set(b) >= {'ques_fwd', 'hist', 'img_feat', 'answer_in', 'answer_out', 'answer_ind'}
