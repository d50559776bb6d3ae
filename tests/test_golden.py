"""Golden vectors (tests/golden/*.npz, made by tests/golden/make_golden.py from the fp64 oracle).
CPU: the oracle still reproduces them bit-for-bit-ish (1e-12).  GPU: the HIP path, fed the SAME stored
parameters / batch / dropout masks through the C ABI, matches them within the fp32 tolerance (1e-4)."""
import glob
import os

import numpy as np
import pytest
import torch

from conftest import ROOT, grad_mismatches, small_params, unexplained_rank_flips
from oracle import visdial_oracle as vo
from visdial_amd.opts import derive

FILES = sorted(f for f in glob.glob(os.path.join(ROOT, 'tests', 'golden', '*__*.npz'))     # <encoder>__<decoder>.npz
               if not os.path.basename(f).startswith(('full__', 'ref_')))     # (full-size: test_full_size_golden.py; executed reference: test_reference_goldens.py)
KW = {'lf-ques': dict(dropout=0.5, imgNorm=1, batchSize=2), 'lf-ques-im-hist': dict(dropout=0.5, imgNorm=1, batchSize=2),
      'hre-ques-im-hist': dict(imgNorm=1, batchSize=2), 'mn-att-ques-im-hist': dict(batchSize=2)}


def load(path):
    enc, dec = os.path.basename(path)[:-4].split('__')
    z = np.load(path)
    get = lambda pre: {k[len(pre):]: z[k] for k in z.files if k.startswith(pre)}
    p = derive(small_params(encoder=enc, decoder=dec, **KW[enc]))
    return enc, dec, p, z, get('param.'), get('batch.'), get('mask.'), get('grad.')


def test_golden_files_present():
    assert len(FILES) == 4


@pytest.mark.parametrize("path", FILES, ids=[os.path.basename(f) for f in FILES])
def test_oracle_reproduces_golden(path):
    enc, dec, p, z, P, batch, masks, grads = load(path)
    P64 = {k: v.astype(np.float64) for k, v in P.items()}
    drop = {k: v.astype(np.float64) for k, v in masks.items()} if masks else None
    r = vo.forward_backward(enc, dec, P64, p, batch, drop)
    assert abs(r['loss'] - float(z['loss'])) < 1e-12 * max(1.0, abs(float(z['loss'])))
    for k, g in grads.items():
        np.testing.assert_allclose(r['grads'][k], g, rtol=1e-10, atol=1e-13, err_msg=k)
    if dec == 'disc':
        ev = vo.forward_backward(enc, dec, P64, p, batch, None, only_forward=True)
        np.testing.assert_array_equal(vo.compute_ranks(ev['scores']), z['eval.ranks'])


@pytest.mark.gpu
@pytest.mark.parametrize("path", FILES, ids=[os.path.basename(f) for f in FILES])
def test_hip_path_matches_golden(path):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from visdial_amd.model import Model
    from visdial_amd import utils
    enc, dec, p, z, P, batch, masks, grads = load(path)
    model = Model(p)
    model.set_parameters_dict(P)
    if masks:
        model.set_dropout_masks(masks)
    else:
        model.wrapper.evaluate()
    model.wrapper.zeroGradParameters()
    loss = model.forwardBackward(batch)
    ref = float(z['loss'])
    assert abs(loss - ref) < 1e-4 * max(1.0, abs(ref))
    g = model.get_gradients_dict()
    bad = grad_mismatches(g, grads)
    assert not bad, bad
    if dec == 'disc':
        model.wrapper.evaluate()
        p['useGt'] = False
        ranks = model.retrieveBatch(batch)
        dev_scores = model.decoder.output.cpu().numpy()
        assert np.abs(dev_scores - z['eval.scores']).max() < 1e-4
        np.testing.assert_array_equal(ranks, vo.compute_ranks(dev_scores))         # bit-exact on device scores
        flipped, unexplained = unexplained_rank_flips(dev_scores, z['eval.scores'])    # and equal to golden up to near ties
        assert not unexplained, unexplained[:10]
        assert (np.asarray(ranks) != z['eval.ranks']).sum() <= 2 * flipped
        p['useGt'] = True
        gt = np.asarray(model.retrieveBatch(batch)).reshape(-1)
        # the GT rank is the oracle's rule on the DEVICE scores, exactly; against the golden ranks it may move only by the near-tie flips above
        np.testing.assert_array_equal(gt, vo.compute_ranks(dev_scores, batch['answer_ind'].reshape(-1) - 1))
        assert np.abs(gt - z['eval.gt_ranks'].reshape(-1)).sum() <= flipped
        m = utils.processRanks(gt, verbose=False)
        want = vo.process_ranks(gt, dev_scores.shape[1])         # utils.lua:131-160 restated, on the same ranks
        np.testing.assert_allclose([m[k] for k in ('r@1', 'r@5', 'r@10', 'medianR', 'meanR', 'meanRR')],
                                   [want[k] for k in ('r@1', 'r@5', 'r@10', 'medianR', 'meanR', 'meanRR')], atol=1e-12)
        if flipped == 0:
            np.testing.assert_allclose([m[k] for k in ('r@1', 'r@5', 'r@10', 'medianR', 'meanR', 'meanRR')], z['eval.metrics'], atol=1e-12)
