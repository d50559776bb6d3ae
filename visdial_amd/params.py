"""Flat parameter / gradient storage -- the analogue of `wrapper:getParameters()` (model.lua:55).

One contiguous fp32 device vector for all parameters (`wrapperW`) and one for all gradients
(`wrapperdW`); modules hold named views into them, weight-shared modules (the embedding used by
question / history / option / answer branches, disc.lua:12, gen.lua:10) hold the SAME view, so
gradients from every user accumulate into one buffer exactly as Torch's shared storages do.

Layout (documented, ours to define -- `.t7` compatibility is a "next" item):
  embed [(V+1) x E] | encoder tensors in plug-in declaration order | decoder tensors
LSTM tensors follow nn.SeqLSTM ([(D+H) x 4H] = [Wx;Wh], gate order i,f,o,g; bias [4H]); Linear
tensors follow nn.Linear ([out x in], [out]).  Every tensor starts on a 16-byte boundary.
"""
import numpy as np
import torch


class ParamSpec(object):
    def __init__(self):
        self.entries = []   # (name, shape, kind)

    def embed(self, name, rows, E):
        self.entries.append((name, (rows, E), 'embed'))

    def lstm(self, name, D, H):
        self.entries.append((name + '.W', (D + H, 4 * H), 'lstm_w'))
        self.entries.append((name + '.b', (4 * H,), 'lstm_b'))

    def linear(self, name, n_in, n_out):
        self.entries.append((name + '.W', (n_out, n_in), 'lin_w'))
        self.entries.append((name + '.b', (n_out,), 'lin_b'))


def init_host(spec, H, seed):
    """Library-default initialisation (model_utils/weight-init.lua is effectively a no-op in the
    reference): SeqLSTM weight ~ N(0, 1/sqrt(D+H)), bias 0 with the forget
    gate at 1; Linear weight and bias ~ U(+-1/sqrt(in)); LookupTable ~ N(0,1) with the pad row zero."""
    rng = np.random.RandomState(seed)
    out = {}
    for name, shape, kind in spec.entries:
        if kind == 'embed':
            w = rng.randn(*shape)
            w[0] = 0
        elif kind == 'lstm_w':
            w = rng.randn(*shape) * (1.0 / np.sqrt(shape[0]))
        elif kind == 'lstm_b':
            w = np.zeros(shape)
            w[H:2 * H] = 1.0
        elif kind == 'lin_w':
            s = 1.0 / np.sqrt(shape[1])
            w = rng.uniform(-s, s, shape)
        elif kind == 'lin_b':
            s = 1.0 / np.sqrt(out[name[:-2] + '.W'].shape[1])
            w = rng.uniform(-s, s, shape)
        else:
            raise AssertionError(kind)
        out[name] = w.astype(np.float32)
    return out


class FlatParams(object):
    """Owns wrapperW / wrapperdW (+ Adam moments) on the device and the named views."""

    def __init__(self, spec, device):
        self.spec = spec
        self.offsets = {}
        off = 0
        for name, shape, _ in spec.entries:
            self.offsets[name] = (off, shape)
            n = int(np.prod(shape))
            off += (n + 3) // 4 * 4          # keep every tensor 16-byte aligned
        self.numel = off
        self.W = torch.zeros(off, dtype=torch.float32, device=device)
        self.dW = torch.zeros(off, dtype=torch.float32, device=device)
        self.w = {n: self.W[o:o + int(np.prod(s))].view(*s) for n, (o, s) in self.offsets.items()}
        self.g = {n: self.dW[o:o + int(np.prod(s))].view(*s) for n, (o, s) in self.offsets.items()}

    def num_parameters(self):
        """count without alignment padding (what getParameters() would report)"""
        return int(sum(np.prod(s) for _, s, _ in self.spec.entries))

    def load_host(self, host_dict):
        for n, (o, s) in self.offsets.items():
            self.w[n].copy_(torch.from_numpy(np.ascontiguousarray(host_dict[n], dtype=np.float32)))

    def to_host(self, which='w'):
        src = self.w if which == 'w' else self.g
        return {n: src[n].detach().cpu().numpy().copy() for n in self.offsets}
