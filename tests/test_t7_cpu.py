"""Torch7 .t7 (de)serialisation round trips (visdial_amd/t7.py) for the checkpoint structure of
train.lua:99-102: {modelW = tensor, optims = {learningRate, t, m, v}, modelParams = table}."""
import struct

import numpy as np
import pytest

from visdial_amd import t7
from visdial_amd.params import ParamSpec


def test_checkpoint_roundtrip(tmp_path):
    rng = np.random.RandomState(0)
    ck = {'modelW': rng.randn(1000).astype(np.float32),
          'optims': {'learningRate': 5e-4, 't': 12, 'm': rng.randn(1000).astype(np.float32), 'v': rng.rand(1000).astype(np.float32)},
          'modelParams': {'encoder': 'lf-ques-im-hist', 'decoder': 'gen', 'rnnHiddenSize': 512, 'useIm': True,
                          'dropout': 0.5, 'nested': {'a': [1, 2, 3]}, 'none': None}}
    p = str(tmp_path / 'm.t7')
    t7.save(p, ck, float_tensor_class='Cuda')
    back = t7.load(p)
    np.testing.assert_array_equal(back['modelW'], ck['modelW'])
    np.testing.assert_array_equal(back['optims']['m'], ck['optims']['m'])
    assert back['optims']['learningRate'] == 5e-4 and back['optims']['t'] == 12
    mp = back['modelParams']
    assert mp['encoder'] == 'lf-ques-im-hist' and mp['useIm'] is True and mp['nested']['a'] == [1, 2, 3] and mp['rnnHiddenSize'] == 512


def test_layout_is_the_documented_one(tmp_path):
    """byte-level check of a tiny object against the format description in the module docstring"""
    p = str(tmp_path / 'x.t7')
    t7.save(p, {'a': np.arange(6, dtype=np.float32).reshape(2, 3)})
    b = open(p, 'rb').read()
    i = lambda o: struct.unpack('<i', b[o:o + 4])[0]
    assert i(0) == 3 and i(4) == 1 and i(8) == 1                      # table, ref 1, one pair
    assert i(12) == 2 and i(16) == 1 and b[20:21] == b'a'             # key: string 'a'
    assert i(21) == 4 and i(25) == 2                                  # value: torch object, ref 2
    assert b[33:36] == b'V 1' and b[40:57] == b'torch.FloatTensor'
    nd = i(57)
    assert nd == 2 and struct.unpack('<qq', b[61:77]) == (2, 3) and struct.unpack('<qq', b[77:93]) == (3, 1)
    np.testing.assert_array_equal(t7.load(p)['a'], np.arange(6, dtype=np.float32).reshape(2, 3))


def test_non_contiguous_and_shared_storage(tmp_path):
    """a hand-written file: two tensors viewing ONE storage with strides/offsets (what getParameters() saves)"""
    p = str(tmp_path / 'v.t7')
    s = np.arange(10, dtype=np.float64)
    with open(p, 'wb') as f:
        w = lambda fmt, *v: f.write(struct.pack('<' + fmt, *v))
        def st(x):
            w('i', len(x)); f.write(x.encode())
        w('i', 3); w('i', 1); w('i', 2)
        w('i', 1); w('d', 1.0)                                        # key 1
        w('i', 4); w('i', 2); st('V 1'); st('torch.DoubleTensor'); w('i', 2); w('qq', 2, 2); w('qq', 1, 2); w('q', 3)
        w('i', 4); w('i', 3); st('V 1'); st('torch.DoubleStorage'); w('q', 10); f.write(s.tobytes())
        w('i', 1); w('d', 2.0)                                        # key 2
        w('i', 4); w('i', 4); st('V 1'); st('torch.DoubleTensor'); w('i', 1); w('q', 4); w('q', 1); w('q', 1)
        w('i', 4); w('i', 3)                                          # the SAME storage by reference
    a, b = t7.load(p)
    np.testing.assert_array_equal(a, [[2, 4], [3, 5]])                # offset 3 (1-based), strides (1, 2)
    np.testing.assert_array_equal(b, [0, 1, 2, 3])


def test_flat_vector_mapping():
    spec = ParamSpec()
    spec.embed('embed', 7, 3); spec.lstm('ques1', 3, 4); spec.linear('fuse', 4, 5)
    named = {n: np.random.RandomState(1).randn(*s).astype(np.float32) for n, s, _ in spec.entries}
    flat = t7.named_to_flat(named, spec.entries)
    assert flat.size == 7 * 3 + 7 * 16 + 16 + 20 + 5
    back = t7.flat_to_named(flat, spec.entries)
    for n in named:
        np.testing.assert_array_equal(back[n], named[n])


def _spec_for(enc, dec, **kw):
    from visdial_amd import encoders, decoders
    from visdial_amd.opts import default_params
    p = default_params(encoder=enc, decoder=dec, vocabSize=20, embedSize=8, rnnHiddenSize=32, imgFeatureSize=12,
                       imgEmbedSize=8, imgSpatialSize=2, commonEmbeddingSize=16, **kw)
    spec = ParamSpec()
    spec.embed('embed', p['vocabSize'] + 1, p['embedSize'])
    encoders.load(enc).declare(p, spec)
    decoders.load(dec).declare(p, spec)
    return spec.entries


def _derived_orders():
    import json
    import os
    return json.load(open(os.path.join(os.path.dirname(__file__), 'golden', 'reference_param_order.json')))


def test_reference_parameter_order_equals_the_executed_reference():
    """t7.reference_order == the order wrapper:getParameters() produced when the reference's encoder / decoder files were
    EXECUTED (tests/golden/make_reference_goldens.py -> reference_param_order.json), for all 11 encoders x both decoders
    and for the numLayers / numAttentionLayers variants recorded there."""
    d = _derived_orders()
    names = lambda entries: [e[0] for e in entries]
    for enc, order in d['encoder'].items():
        for dec in ('disc', 'gen'):
            e = _spec_for(enc, dec)
            ref = names(t7.reference_order(enc, e))
            assert ref == order + d['decoder'][dec], (enc, dec)
            assert sorted(ref) == sorted(names(e))
            # a flat vector written in reference order comes back tensor by tensor
            named = {n: np.random.RandomState(len(n)).randn(*s).astype(np.float32) for n, s, _ in e}
            back = t7.flat_to_named(t7.named_to_flat(named, e, enc), e, enc)
            for n in named:
                np.testing.assert_array_equal(back[n], named[n])
    assert len(d['encoder']) == 11 and len(d['variants']) >= 5
    for v in d['variants']:
        e = _spec_for(v['encoder'], v['decoder'], **v['params'])
        assert names(t7.reference_order(v['encoder'], e)) == v['order'], v
    # the documented landmarks: hre*-im puts the image Linear right behind the embedding, hre-ques-hist runs the QUESTION LSTMs
    # first, the nngraph att encoders start with the image tower
    assert d['encoder']['hre-ques-im-hist'][:3] == ['embed', 'img_embed.W', 'img_embed.b']
    assert d['encoder']['hre-ques-hist'][1] == 'ques1.W' and d['encoder']['mn-att-ques-im-hist'][0] == 'img_proj.W'
    assert t7.order_status('mn-att-ques-im-hist') == 'derived' and t7.order_status('hre-ques-hist') == 'source'
    with pytest.raises(ValueError):                       # the element count is always checked
        e = _spec_for('mn-att-ques-im-hist', 'disc')
        t7.flat_to_named(np.zeros(sum(int(np.prod(s)) for _, s, _ in e) + 1, np.float32), e, 'mn-att-ques-im-hist')


class _FakeModel(object):
    """the three things checkpoint.save_t7 / restore_weights use of a host model"""

    def __init__(self, encoder, named):
        self.params = {'encoder': encoder}
        self.optims = {'learningRate': 7e-4}
        self.named = named

    def _entries(self):
        return [(k, v.shape, 'lin_w') for k, v in self.named.items()]

    def get_parameters_dict(self):
        return self.named

    def set_parameters_dict(self, d):
        self.named = {k: np.array(v) for k, v in d.items()}

    def load_flat_parameters(self, w):
        self.set_parameters_dict(t7.flat_to_named(w, self._entries(), self.params['encoder']))


def _fake_named(enc, dec='disc'):
    rng = np.random.RandomState(3)
    return {n: rng.randn(*s).astype(np.float32) for n, s, _ in _spec_for(enc, dec)}


def test_cli_checkpoint_t7_roundtrip_in_reference_order(tmp_path, capsys):
    """train.py's model_epoch_%d.t7 / model_final.t7 (train.lua:99-102,120-121): the flat vector is in the reference's
    getParameters() order for EVERY encoder (t7.reference_order); loading an nngraph encoder says the order is derived."""
    from visdial_amd import checkpoint
    for enc in ('hre-ques-im-hist', 'hre-ques-hist', 'mn-att-ques-im-hist'):
        named = _fake_named(enc)
        m = _FakeModel(enc, dict(named))
        p = str(tmp_path / (enc + '.t7'))
        checkpoint.save_t7(p, m, {'encoder': enc, 'decoder': 'disc', 'rnnHiddenSize': 32})
        ck = checkpoint.load_checkpoint(p)
        assert ck.get('vdLayout') == 'reference' and abs(ck['optims']['learningRate'] - 7e-4) < 1e-12
        first = {'hre-ques-im-hist': ('embed', 'img_embed.W'), 'hre-ques-hist': ('embed', 'ques1.W'),
                 'mn-att-ques-im-hist': ('img_proj.W', 'img_proj.b')}[enc]
        o = 0
        for n in first:
            np.testing.assert_array_equal(ck['modelW'][o:o + named[n].size], named[n].reshape(-1))
            o += named[n].size
        m2 = _FakeModel(enc, {k: np.zeros_like(v) for k, v in named.items()})
        capsys.readouterr()
        checkpoint.restore_weights(m2, ck)
        err = capsys.readouterr().err
        assert ('derived' in err) == (enc == 'mn-att-ques-im-hist') and 'no vdLayout marker' not in err
        for k in named:
            np.testing.assert_array_equal(m2.named[k], named[k])
    # a file an EARLIER version of this repo wrote for an nngraph encoder (declaration order + marker) still loads
    ck['modelW'] = t7.named_to_flat(named, m._entries(), None)
    ck['vdLayout'] = 'declaration'
    m3 = _FakeModel(enc, {k: np.zeros_like(v) for k, v in named.items()})
    checkpoint.restore_weights(m3, ck)
    for k in named:
        np.testing.assert_array_equal(m3.named[k], named[k])


def test_hand_assembled_torch7_checkpoint(tmp_path):
    """A checkpoint assembled BYTE BY BYTE here, the way Torch7's File:writeObject lays out what train.lua:99-102 saves on a GPU
    ({modelW = CudaTensor viewing a storage, optims = {learningRate}, modelParams = {...}}; torch7/File.lua, binary mode) --
    independent of t7.save.  It loads through checkpoint.load_checkpoint / restore_weights into the named tensors in the
    reference's getParameters() order."""
    from visdial_amd import checkpoint
    enc = 'mn-ques-im-hist'
    named = _fake_named(enc)
    order = _derived_orders()['encoder'][enc] + ['opt.W', 'opt.b']
    flat = np.concatenate([named[n].reshape(-1) for n in order]).astype(np.float32)
    p = str(tmp_path / 'hand.t7')
    with open(p, 'wb') as f:
        w = lambda fmt, *v: f.write(struct.pack('<' + fmt, *v))

        def st(x):
            w('i', 2); w('i', len(x)); f.write(x.encode())

        def raw(x):
            w('i', len(x)); f.write(x.encode())
        w('i', 3); w('i', 1); w('i', 3)                                   # table, ref 1, three pairs
        st('modelW')
        w('i', 4); w('i', 2); raw('V 1'); raw('torch.CudaTensor')         # torch object, ref 2
        w('i', 1); w('q', flat.size); w('q', 1); w('q', 1)                # nDim, size, stride, storageOffset (1-based)
        w('i', 4); w('i', 3); raw('V 1'); raw('torch.CudaStorage'); w('q', flat.size); f.write(flat.tobytes())
        st('optims')
        w('i', 3); w('i', 4); w('i', 1); st('learningRate'); w('i', 1); w('d', 2.5e-4)
        st('modelParams')
        w('i', 3); w('i', 5); w('i', 3)
        st('encoder'); st(enc)
        st('rnnHiddenSize'); w('i', 1); w('d', 32.0)
        st('useIm'); w('i', 5); w('i', 1)
    ck = checkpoint.load_checkpoint(p)
    assert ck['modelParams'] == {'encoder': enc, 'rnnHiddenSize': 32, 'useIm': True} and ck['optims']['learningRate'] == 2.5e-4
    m = _FakeModel(enc, {k: np.zeros_like(v) for k, v in named.items()})
    checkpoint.restore_weights(m, ck)
    for k in named:
        np.testing.assert_array_equal(m.named[k], named[k])
    # the same bytes come out of t7.save for the same object (so a file this repo writes is laid out like the hand-assembled one)
    q = str(tmp_path / 'saved.t7')
    t7.save(q, {'modelW': flat, 'optims': {'learningRate': 2.5e-4}, 'modelParams': {'encoder': enc, 'rnnHiddenSize': 32, 'useIm': True}},
            float_tensor_class='Cuda')
    assert open(q, 'rb').read() == open(p, 'rb').read()


def test_param_order_escape_hatch(tmp_path, capsys):
    """-paramOrder: a user with a real Torch7 runs lua/dump_param_order.lua on a checkpoint and hands the JSON to
    the loaders.  (a) a dump that agrees with this repo's table in every tensor size keeps the table and names the same-size groups it
    cannot tell apart; (b) a dump that contradicts the table wins; (c) an explicit name list is followed verbatim; (d) a file without a
    layout marker is announced as such for the encoders whose order changed between rounds; (e) the fresh-initialisation forget-bias
    pattern is reported."""
    import json
    from visdial_amd import checkpoint
    enc = 'mn-att-ques-im-hist'
    named = _fake_named(enc)
    for k in named:                                     # a fresh reference init: forget-gate quarter of every LSTM bias = 1
        if k.endswith('.b') and named[k[:-2] + '.W'].shape[0] > named[k].size // 4 and named[k[:-2] + '.W'].shape[-1] == named[k].size:
            named[k][named[k].size // 4:named[k].size // 2] = 1.0
    m = _FakeModel(enc, dict(named))
    entries = m._entries()
    table = t7.reference_order(enc, entries)

    def dump_of(order):                                 # what lua/dump_param_order.lua prints for a flat vector laid out in `order`
        rows, o = [], 0
        for i, (n, shape, _) in enumerate(order):
            rows.append({'module': i + 1, 'type': 'nn.Linear', 'field': 'weight', 'offset': o, 'numel': int(np.prod(shape)), 'rows': int(shape[0])})
            o += int(np.prod(shape))
        return {'encoder': enc, 'decoder': 'disc', 'total': o, 'tensors': rows}
    flat = t7.named_to_flat(named, entries, enc)
    ck = {'modelW': flat, '_flat_reference_layout': True}
    # (a) + (d) + (e)
    pa = str(tmp_path / 'agree.json')
    json.dump(dump_of(table), open(pa, 'w'))
    m2 = _FakeModel(enc, {k: np.zeros_like(v) for k, v in named.items()})
    capsys.readouterr()
    checkpoint.restore_weights(m2, ck, pa)
    err = capsys.readouterr().err
    assert 'agrees with the Torch7 dump' in err and "'ques_common.W'" in err and 'forget-gate-bias-1 pattern' in err
    for k in named:
        np.testing.assert_array_equal(m2.named[k], named[k])
    capsys.readouterr()
    checkpoint.restore_weights(_FakeModel(enc, dict(named)), ck)
    assert 'no vdLayout marker' in capsys.readouterr().err
    # (b) the real order differs from the table: embed first
    real = [e for e in table if e[0] == 'embed'] + [e for e in table if e[0] != 'embed']
    flat_real = np.concatenate([named[n].reshape(-1) for n, _, _ in real])
    pb = str(tmp_path / 'contradict.json')
    json.dump(dump_of(real), open(pb, 'w'))
    m3 = _FakeModel(enc, {k: np.zeros_like(v) for k, v in named.items()})
    capsys.readouterr()
    checkpoint.restore_weights(m3, {'modelW': flat_real, '_flat_reference_layout': True}, pb)
    assert 'CONTRADICTS the table from position 0' in capsys.readouterr().err
    for k in named:
        np.testing.assert_array_equal(m3.named[k], named[k])
    # (c) explicit list, including a permutation of two same-size tensors that no dump could express
    names = [e[0] for e in table]
    i, j = names.index('ques_common.W'), names.index('att.W') if named['att.W'].size == named['ques_common.W'].size else names.index('mn1.W')
    names[i], names[j] = names[j], names[i]
    flat_c = np.concatenate([named[n].reshape(-1) for n in names])
    pc = str(tmp_path / 'explicit.json')
    json.dump({'order': names}, open(pc, 'w'))
    m4 = _FakeModel(enc, {k: np.zeros_like(v) for k, v in named.items()})
    checkpoint.restore_weights(m4, {'modelW': flat_c, '_flat_reference_layout': True}, pc)
    for k in named:
        np.testing.assert_array_equal(m4.named[k], named[k])
    with pytest.raises(ValueError):
        t7.resolve_order(entries, enc, {'order': names[:-1]})
    # a mis-split fresh init is flagged: read a declaration-order vector with the reference table
    bad = _FakeModel(enc, {k: np.zeros_like(v) for k, v in named.items()})
    capsys.readouterr()
    checkpoint.restore_weights(bad, {'modelW': t7.named_to_flat(named, entries, None), '_flat_reference_layout': True, 'vdLayout': 'reference'})
    assert 'CHECK the parameter order' in capsys.readouterr().err or any((bad.named[k] != named[k]).any() for k in named)
