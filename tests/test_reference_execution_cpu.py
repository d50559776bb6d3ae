"""Re-EXECUTE the reference's Lua (tests/luavm + tests/luavm/nn7.py) where /root/reference exists -- the build container, never the GPU box --
and check that the committed fixtures (tests/golden/ref_*.npz) are what the committed generator scripts produce: one model pair
(make_reference_goldens.py), the dataloader (make_reference_dataloader_golden.py) and the unedited train.lua / evaluate.lua / generate.lua
run (make_reference_train_golden.py).  Skipped where the reference is absent."""
import os
import sys

import numpy as np
import pytest

from conftest import ROOT

REF = '/root/reference'
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, 'encoders')), reason='the reference checkout is not on this machine')
sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))


def same(rec, path, skip=()):
    z = np.load(path)
    assert set(rec) == set(z.files), sorted(set(rec) ^ set(z.files))[:8]
    for k in z.files:
        if k in skip:
            continue
        a, b = np.asarray(rec[k]), z[k]
        assert a.shape == b.shape and a.dtype == b.dtype, k
        if a.dtype.kind in 'fc':
            np.testing.assert_array_equal(a, b, err_msg=k)          # the generators are deterministic: bit for bit
        else:
            assert (a == b).all(), k


PAIRS = [(e, 'disc') for e in ('lf-ques', 'lf-ques-im', 'lf-ques-hist', 'lf-ques-im-hist', 'lf-att-ques-im-hist', 'hre-ques-hist', 'hre-ques-im-hist',
                               'hrea-ques-im-hist', 'mn-ques-hist', 'mn-ques-im-hist', 'mn-att-ques-im-hist')] + \
        [('lf-ques', 'gen'), ('lf-ques-im-hist', 'gen'), ('mn-att-ques-im-hist', 'gen')]


@pytest.mark.parametrize("enc,dec", PAIRS, ids=['%s+%s' % p for p in PAIRS])
def test_model_pair_fixture_is_reproduced(enc, dec):
    """model.lua + encoders/<enc>.lua + decoders/<dec>.lua executed: forwardBackward (evaluate / training), trainIteration, retrieveBatch,
    [generateAnswers, the split loops, three iterations] -- every assertion of the generator against the oracle runs again here"""
    import make_reference_goldens as g
    p, rec, order, report = g.run_pair(enc, dec)
    assert report[0].startswith('evaluate:') and any('trainIteration' in r for r in report)
    same(rec, os.path.join(ROOT, 'tests', 'golden', 'ref__%s__%s.npz' % (enc, dec)))


def test_parameter_order_file_is_reproduced():
    import json
    import make_reference_goldens as g
    d = json.load(open(os.path.join(ROOT, 'tests', 'golden', 'reference_param_order.json')))
    for v in g.variant_orders():
        assert v in d['variants']
    for enc in ('hre-ques-hist', 'mn-att-ques-im-hist'):
        assert g.run_pair(enc, 'disc', order_only=True)[2] == d['encoder'][enc] + d['decoder']['disc']


@pytest.mark.parametrize("case", ['lf-ques-im-hist', 'mn-att-concat', 'lf-ques', 'prepro'])
def test_dataloader_fixture_is_reproduced(case):
    import make_reference_dataloader_golden as g
    rec = g.run_case(case, g.CASES[case])
    same(rec, os.path.join(ROOT, 'tests', 'golden', 'ref_dataloader__%s.npz' % case), skip=('info.json', 'opt.json'))


@pytest.mark.parametrize("enc,dec", [('lf-ques-im-hist', 'gen'), ('mn-att-ques-im-hist', 'disc')])
def test_train_evaluate_generate_fixture_is_reproduced(enc, dec):
    """the unedited train.lua, evaluate.lua [, generate.lua] end to end"""
    import make_reference_train_golden as g
    rec, worst = g.run_case(enc, dec)
    assert worst < 1e-11
    same(rec, os.path.join(ROOT, 'tests', 'golden', 'ref_train__%s__%s.npz' % (enc, dec)), skip=('info.json', 'opt.json'))
