"""Torch7 binary serialisation (`torch.save` / `torch.load` of the reference's checkpoints:
train.lua:99-102,120-121 write {modelW, optims, modelParams}; evaluate.lua:58 and generate.lua:52 read it).

Format restated from Torch7's File:writeObject/readObject (torch7/File.lua, binary mode, 64-bit longs) --
third-party code that is NOT in /root/reference; there is no .t7 file or Torch7 runtime here, so this module is
validated by round trips only and is UNPINNED against real Torch7 output:
  object   := int32 type ; payload
  type     : 0 nil | 1 number (float64) | 2 string (int32 len, bytes) | 3 table | 4 torch object | 5 boolean (int32)
  table    := int32 ref-index ; [first time:] int32 n ; n x (key object, value object)
  torch    := int32 ref-index ; [first time:] string "V 1" ; string className ; class payload
  Tensor   := int32 nDim ; int64 size[nDim] ; int64 stride[nDim] ; int64 storageOffset(1-based) ; storage object
  Storage  := int64 n ; n raw elements
Lua tables with keys 1..n come back as Python lists, everything else as dicts."""
import struct

import numpy as np

_DT = {'Float': np.float32, 'Double': np.float64, 'Long': np.int64, 'Int': np.int32, 'Short': np.int16,
       'Byte': np.uint8, 'Char': np.int8, 'Cuda': np.float32, 'CudaLong': np.int64, 'CudaByte': np.uint8}


class TorchObject(dict):
    """a torch class instance we do not model (e.g. an nn module): className + its field table"""

    def __init__(self, cls, fields=None):
        dict.__init__(self, fields or {})
        self.torch_class = cls


class _Reader(object):
    def __init__(self, f):
        self.f, self.memo = f, {}

    def _u(self, fmt):
        n = struct.calcsize(fmt)
        b = self.f.read(n)
        if len(b) != n:
            raise EOFError('truncated .t7 file')
        return struct.unpack('<' + fmt, b)[0]

    def _str(self):
        return self.f.read(self._u('i')).decode('latin-1')

    def obj(self):
        t = self._u('i')
        if t == 0:
            return None
        if t == 1:
            v = self._u('d')
            return int(v) if v == int(v) and abs(v) < 2 ** 53 else v
        if t == 2:
            return self._str()
        if t == 5:
            return self._u('i') == 1
        if t in (3, 4):
            idx = self._u('i')
            if idx in self.memo:
                return self.memo[idx]
            if t == 3:
                out = {}
                self.memo[idx] = out
                n = self._u('i')
                for _ in range(n):
                    k = self.obj()
                    out[k] = self.obj()
                if out and all(isinstance(k, int) for k in out) and sorted(out) == list(range(1, len(out) + 1)):
                    lst = [out[i] for i in range(1, len(out) + 1)]
                    self.memo[idx] = lst
                    return lst
                return out
            ver = self._str()
            cls = self._str() if ver.startswith('V ') else ver
            kind = cls.split('.')[-1]
            if kind.endswith('Storage') and kind[:-7] in _DT:
                n = self._u('q')
                dt = np.dtype(_DT[kind[:-7]])
                arr = np.frombuffer(self.f.read(n * dt.itemsize), dtype=dt).copy()
                self.memo[idx] = arr
                return arr
            if kind.endswith('Tensor') and kind[:-6] in _DT:
                nd = self._u('i')
                size = [self._u('q') for _ in range(nd)]
                stride = [self._u('q') for _ in range(nd)]
                off = self._u('q') - 1
                storage = self.obj()
                if storage is None or nd == 0:
                    arr = np.zeros(size, _DT[kind[:-6]])
                else:
                    arr = np.lib.stride_tricks.as_strided(storage[off:], shape=size,
                                                          strides=[s * storage.itemsize for s in stride]).copy()
                self.memo[idx] = arr
                return arr
            o = TorchObject(cls)
            self.memo[idx] = o
            fields = self.obj()
            if isinstance(fields, dict):
                o.update(fields)
            return o
        raise ValueError('unsupported .t7 object type %d' % t)


def load(path):
    with open(path, 'rb') as f:
        return _Reader(f).obj()


class _Writer(object):
    def __init__(self, f, tensor_class):
        self.f, self.next, self.cls = f, 1, tensor_class

    def _p(self, fmt, v):
        self.f.write(struct.pack('<' + fmt, v))

    def _str(self, s):
        b = s.encode('latin-1')
        self._p('i', len(b))
        self.f.write(b)

    def _idx(self):
        i = self.next
        self.next += 1
        self._p('i', i)

    def obj(self, o):
        if o is None:
            return self._p('i', 0)
        if isinstance(o, (bool, np.bool_)):
            self._p('i', 5)
            return self._p('i', 1 if o else 0)
        if isinstance(o, (int, float, np.integer, np.floating)):
            self._p('i', 1)
            return self._p('d', float(o))
        if isinstance(o, str):
            self._p('i', 2)
            return self._str(o)
        if isinstance(o, np.ndarray):
            name = {np.dtype(np.float32): 'Float', np.dtype(np.float64): 'Double', np.dtype(np.int64): 'Long',
                    np.dtype(np.int32): 'Int', np.dtype(np.uint8): 'Byte'}[o.dtype]
            if self.cls and o.dtype == np.float32:
                name = self.cls
            a = np.ascontiguousarray(o)
            self._p('i', 4); self._idx(); self._str('V 1'); self._str('torch.%sTensor' % name)
            self._p('i', a.ndim)
            for s in a.shape:
                self._p('q', s)
            st = [int(np.prod(a.shape[i + 1:])) for i in range(a.ndim)]
            for s in st:
                self._p('q', s)
            self._p('q', 1)
            self._p('i', 4); self._idx(); self._str('V 1'); self._str('torch.%sStorage' % name)
            self._p('q', a.size)
            return self.f.write(a.tobytes())
        if isinstance(o, (list, tuple)):
            o = {i + 1: v for i, v in enumerate(o)}
        if isinstance(o, dict):
            self._p('i', 3); self._idx()
            self._p('i', len(o))
            for k, v in o.items():
                self.obj(k)
                self.obj(v)
            return
        raise TypeError('cannot serialise %r to .t7' % type(o))


def save(path, obj, float_tensor_class=None):
    """float_tensor_class='Cuda' writes fp32 arrays as torch.CudaTensor (what train.lua saves on a GPU)."""
    with open(path, 'wb') as f:
        _Writer(f, float_tensor_class).obj(obj)


# ---------------------------------------------------------------------------------------------------------
# flat parameter vector <-> named tensors
# Torch7's getParameters() flattens parameter storages in depth-first module order.  For the
# encoders built from nn.Sequential / nn.ConcatTable that order can be read off the reference file:
#   lf-*    : wordBranch(embed, ques LSTMs) | [image: no parameters] | histBranch(hist LSTMs) | fuse Linear
#             (encoders/lf-ques-im-hist.lua:12-58) -- equals this repo's declaration order;
#   hre-*   : concat = wordBranch(embed), imageBranch(Linear F->imgEmbed), histBranch(hist LSTMs)
#             (encoders/hre-ques-im-hist.lua:56-60), then the question LSTMs (:72-80), [hrea: the two Linear(H,1),
#             hrea-ques-im-hist.lua:88-95], then the dialog LSTM (:92) -- the image Linear comes BEFORE the history
#             LSTMs, unlike this repo's declaration order (hist, img_embed, ques, ...).
# The nngraph encoders (mn-*, lf-att-*) are flattened in nngraph's internal node order, which cannot be derived
# without nngraph: UNVERIFIED, refused unless the caller insists.
VERIFIED_ORDER = ('lf-ques', 'lf-ques-im', 'lf-ques-hist', 'lf-ques-im-hist', 'hre-ques-hist', 'hre-ques-im-hist',
                  'hrea-ques-im-hist')


def reference_order(encoder, spec_entries):
    """spec entries re-ordered to the reference's getParameters() order for `encoder`"""
    entries = list(spec_entries)
    if encoder is not None and encoder.startswith('hre'):
        img = [e for e in entries if e[0].startswith('img_embed.')]
        rest = [e for e in entries if not e[0].startswith('img_embed.')]
        at = 1 if rest and rest[0][0] == 'embed' else 0
        entries = rest[:at] + img + rest[at:]
    return entries


def flat_to_named(modelW, spec_entries, encoder=None, allow_unverified=False):
    """Split a reference-style flat vector (getParameters(): tensors back to back, NO alignment padding) into
    this repo's named tensors, using the reference's parameter order for `encoder` (reference_order).  encoder=None
    keeps this repo's own declaration order (checkpoints written by this repo's named_to_flat with encoder=None)."""
    if encoder is not None and encoder not in VERIFIED_ORDER and not allow_unverified:
        raise ValueError("the getParameters() order of the nngraph encoder '%s' cannot be derived without nngraph "
                         "(SURVEY.md App. A7): refusing to load a reference flat vector (it would load silently "
                         "scrambled); pass allow_unverified=True to assume declaration order" % encoder)
    out, o = {}, 0
    for name, shape, _ in reference_order(encoder, spec_entries):
        n = int(np.prod(shape))
        out[name] = np.asarray(modelW[o:o + n], np.float32).reshape(shape)
        o += n
    if o != len(modelW):
        raise ValueError('checkpoint holds %d parameters, the model declares %d' % (len(modelW), o))
    return out


def named_to_flat(named, spec_entries, encoder=None):
    return np.concatenate([np.asarray(named[n], np.float32).reshape(-1)
                           for n, _, _ in reference_order(encoder, spec_entries)])
