"""Torch7 .t7 (de)serialisation round trips (visdial_amd/t7.py) for the checkpoint structure of
train.lua:99-102: {modelW = tensor, optims = {learningRate, t, m, v}, modelParams = table}."""
import struct

import numpy as np

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


def test_reference_parameter_order_per_encoder_family():
    """getParameters() order of a reference checkpoint: lf-* = declaration order; hre-* puts the
    image Linear BEFORE the history LSTMs (concat = wordBranch, imageBranch, histBranch: hre-ques-im-hist.lua:56-60);
    nngraph encoders (mn-*, lf-att) are refused because their order cannot be derived."""
    import pytest
    from visdial_amd import encoders, decoders
    from visdial_amd.opts import default_params

    def spec_for(enc, dec):
        p = default_params(encoder=enc, decoder=dec, vocabSize=20, embedSize=8, rnnHiddenSize=32, imgFeatureSize=12,
                           imgEmbedSize=8, imgSpatialSize=2, commonEmbeddingSize=16)
        spec = ParamSpec()
        spec.embed('embed', p['vocabSize'] + 1, p['embedSize'])
        encoders.load(enc).declare(p, spec)
        decoders.load(dec).declare(p, spec)
        return spec.entries

    names = lambda entries: [e[0] for e in entries]
    e = spec_for('lf-ques-im-hist', 'gen')
    assert names(t7.reference_order('lf-ques-im-hist', e)) == names(e)
    assert names(e)[:3] == ['embed', 'ques1.W', 'ques1.b'] and 'fuse.W' in names(e)
    for enc in ('hre-ques-im-hist', 'hrea-ques-im-hist'):
        e = spec_for(enc, 'disc')
        ref = names(t7.reference_order(enc, e))
        assert ref[:3] == ['embed', 'img_embed.W', 'img_embed.b'] and ref[3] == 'hist1.W'
        assert sorted(ref) == sorted(names(e)) and ref.index('ques1.W') > ref.index('hist2.b')
        # a flat vector written in reference order comes back tensor by tensor
        named = {n: np.random.RandomState(len(n)).randn(*s).astype(np.float32) for n, s, _ in e}
        flat = t7.named_to_flat(named, e, enc)
        back = t7.flat_to_named(flat, e, enc)
        for n in named:
            np.testing.assert_array_equal(back[n], named[n])
        # ... and differs from declaration order exactly where the image Linear moved
        assert not np.array_equal(flat, t7.named_to_flat(named, e))
    e = spec_for('hre-ques-hist', 'disc')
    assert names(t7.reference_order('hre-ques-hist', e)) == names(e)
    for enc in ('mn-att-ques-im-hist', 'mn-ques-hist', 'lf-att-ques-im-hist'):
        e = spec_for(enc, 'disc')
        flat = np.zeros(sum(int(np.prod(s)) for _, s, _ in e), np.float32)
        with pytest.raises(ValueError):
            t7.flat_to_named(flat, e, enc)
        t7.flat_to_named(flat, e, enc, allow_unverified=True)


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

    def load_flat_parameters(self, w, allow_unverified=False):
        self.set_parameters_dict(t7.flat_to_named(w, self._entries(), self.params['encoder'], allow_unverified))


def test_cli_checkpoint_t7_roundtrip_marks_unverified_layouts(tmp_path):
    """train.py's model_epoch_%d.t7 / model_final.t7 (train.lua:99-102,120-121): Sequential encoders are written in the
    reference's getParameters() order; nngraph encoders in this library's declaration order WITH a marker, and
    restore_weights follows the marker instead of guessing (a Torch7-written file without it stays refused)."""
    import pytest
    from visdial_amd import checkpoint
    rng = np.random.RandomState(3)
    named = {'embed': rng.randn(5, 4).astype(np.float32), 'hist1.W': rng.randn(6, 8).astype(np.float32),
             'img_embed.W': rng.randn(3, 4).astype(np.float32), 'ques1.W': rng.randn(6, 8).astype(np.float32)}
    for enc, marked in (('hre-ques-im-hist', False), ('mn-att-ques-im-hist', True)):
        m = _FakeModel(enc, dict(named))
        p = str(tmp_path / (enc + '.t7'))
        checkpoint.save_t7(p, m, {'encoder': enc, 'decoder': 'disc', 'rnnHiddenSize': 8})
        ck = checkpoint.load_checkpoint(p)
        assert ('vdLayout' in ck) == marked and abs(ck['optims']['learningRate'] - 7e-4) < 1e-12
        if not marked:     # reference order: the image Linear sits between embed and the history LSTM
            np.testing.assert_array_equal(ck['modelW'][20:32], named['img_embed.W'].reshape(-1))
        m2 = _FakeModel(enc, {k: np.zeros_like(v) for k, v in named.items()})
        checkpoint.restore_weights(m2, ck)
        for k in named:
            np.testing.assert_array_equal(m2.named[k], named[k])
    # the same vector WITHOUT the marker is what Torch7 would have written: refused unless the caller insists
    ck.pop('vdLayout')
    with pytest.raises(ValueError):
        checkpoint.restore_weights(m2, ck)
    checkpoint.restore_weights(m2, ck, allow_unverified=True)
