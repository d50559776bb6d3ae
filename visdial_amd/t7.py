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
import re
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
# Torch7's getParameters() flattens the parameter storages in module order, every shared storage at its first
# occurrence.  The orders below were obtained by EXECUTING the reference's encoder / decoder files
# (tests/golden/make_reference_goldens.py: the reference's Lua on tests/luavm + a numpy restatement of nn / nngraph /
# rnn) and reading the offsets of every tensor in wrapper:getParameters(); tests/golden/reference_param_order.json
# holds the result and tests/test_reference_goldens.py keeps this table equal to it.
#   lf-*    : wordBranch(embed, ques LSTMs) | histBranch(hist LSTMs) | fuse Linear        (lf-ques-im-hist.lua:12-58)
#   hre-ques-hist : embed, ques LSTMs, hist LSTMs, dialog LSTM                                (hre-ques-hist.lua)
#   hre(a)-ques-im-hist : embed, image Linear, hist LSTMs (hre-ques-im-hist.lua:56-60), ques LSTMs (:72-80),
#             [hrea: the two Linear(H,1), hrea-ques-im-hist.lua:88-95], dialog LSTM (:92)
#   mn-* / lf-att-* (nngraph): gModule keeps its modules in forward-node order = depth-first post-order from the
#             output node over each node's inputs in declaration order.  mn-att with L attention hops: the hop-L
#             img_common is reached FIRST (img_common<L> .. img_common2, img_common), the (ques_common, att)
#             pairs follow in hop order.
# STATUS: for the Sequential/ConcatTable encoders the order can also be read off the source.  For the four nngraph
# encoders it rests on the restated nngraph (graph.topsort of the reversed graph) -- DERIVED, NOT VERIFIED against a
# Torch7-written checkpoint (none exists offline); loaders say so (order_status) and the element count is checked.
NNGRAPH_ENCODERS = ('lf-att-ques-im-hist', 'mn-ques-hist', 'mn-ques-im-hist', 'mn-att-ques-im-hist')
_STEMS = {
    'lf-ques': ('embed', 'ques*', 'fuse'),
    'lf-ques-im': ('embed', 'ques*', 'fuse'),
    'lf-ques-hist': ('embed', 'ques*', 'hist*', 'fuse'),
    'lf-ques-im-hist': ('embed', 'ques*', 'hist*', 'fuse'),
    'lf-att-ques-im-hist': ('img_proj', 'img_common', 'embed', 'ques*', 'hist*', 'qh', 'ques_common', 'att', 'out'),
    'hre-ques-hist': ('embed', 'ques*', 'hist*', 'dialog'),
    'hre-ques-im-hist': ('embed', 'img_embed', 'hist*', 'ques*', 'dialog'),
    'hrea-ques-im-hist': ('embed', 'img_embed', 'hist*', 'ques*', 'att_q', 'att_h', 'dialog'),
    'mn-ques-hist': ('embed', 'ques*', 'hist*', 'mn1', 'mn2'),
    'mn-ques-im-hist': ('embed', 'ques*', 'qi', 'hist*', 'mn1', 'mn2'),
    'mn-att-ques-im-hist': ('img_proj', 'img_common#rev', 'embed', 'ques*', 'hist*', 'mn1', 'mn2', 'ques_common+att#', 'out'),
}


def order_status(encoder):
    """'source' (order readable off the Sequential/ConcatTable source and reproduced by executing it) or 'derived'
    (nngraph encoders: obtained by executing the reference file on a restated nngraph, not verified on a Torch7 file)"""
    return 'derived' if encoder in NNGRAPH_ENCODERS else 'source'


def reference_order(encoder, spec_entries):
    """spec entries re-ordered to the reference's getParameters() order for `encoder`; the decoder's tensors (everything
    the encoder table does not name) follow in declaration order (decoders/disc.lua, decoders/gen.lua)"""
    entries = list(spec_entries)
    if encoder is None:
        return entries
    stem = lambda n: n.rsplit('.', 1)[0]
    stems = []
    for e in entries:
        if stem(e[0]) not in stems:
            stems.append(stem(e[0]))
    layered = lambda pre: sorted((s for s in stems if re.fullmatch(pre + r'\d+', s)), key=lambda s: int(s[len(pre):]))
    hops = lambda pre: [pre] + layered(pre)                   # hop 1 has no suffix, hop i > 1 is <name><i>
    want = []
    for t in _STEMS[encoder]:
        if t.endswith('*'):
            want += layered(t[:-1])
        elif t == 'img_common#rev':
            want += list(reversed(hops('img_common')))
        elif t == 'ques_common+att#':
            for q, a in zip(hops('ques_common'), hops('att')):
                want += [q, a]
        else:
            want.append(t)
    missing = [s for s in want if s not in stems]
    if missing:
        raise ValueError("reference_order(%s): the model declares no tensor named %s" % (encoder, missing))
    want += [s for s in stems if s not in want]              # the decoder
    return [e for s in want for e in entries if stem(e[0]) == s]


def resolve_order(spec_entries, encoder=None, param_order=None, log=None):
    """The flat-vector order to use: this repo's table (reference_order) unless `param_order` overrides it.
      None / '' / 'reference'   the table (Torch7 files, files written by this repo with vdLayout = 'reference')
      'declaration'             the library's own declaration order (files with vdLayout = 'declaration': earlier writers of this repo)
      {'order': [names]}        an explicit list of tensor names in flat order
      {'tensors': [...]}        the output of lua/dump_param_order.lua run under a real Torch7: offsets, element counts and module types of
                                every parameter tensor in flat order.  The table's order is CHECKED against it: where the element-count
                                sequences agree the table stands (tensors of equal size cannot be told apart by a dump: they are reported);
                                where they differ the dump wins -- every position takes the first unused tensor of the dumped size.
    `log(str)` receives what was found."""
    say = log or (lambda m: None)
    entries = list(spec_entries)
    if param_order in (None, '', 'reference'):
        return reference_order(encoder, entries)
    if param_order == 'declaration':
        return entries
    if isinstance(param_order, str):
        import json
        with open(param_order) as f:
            param_order = json.load(f)
    by_name = {e[0]: e for e in entries}
    if 'order' in param_order:
        names = list(param_order['order'])
        if sorted(names) != sorted(by_name):
            raise ValueError('paramOrder: the list must name every tensor exactly once; missing %s, unknown %s' % (
                sorted(set(by_name) - set(names)), sorted(set(names) - set(by_name))))
        return [by_name[n] for n in names]
    dump = sorted(param_order['tensors'], key=lambda t: t['offset'])
    table = reference_order(encoder, entries)
    numel = lambda e: int(np.prod(e[1]))
    if len(dump) != len(table) or sum(int(t['numel']) for t in dump) != sum(numel(e) for e in table):
        raise ValueError('paramOrder: the dump lists %d tensors / %d parameters, the model declares %d / %d (different options?)' % (
            len(dump), sum(int(t['numel']) for t in dump), len(table), sum(numel(e) for e in table)))
    if [int(t['numel']) for t in dump] == [numel(e) for e in table]:
        groups = {}
        for e in table:
            groups.setdefault(numel(e), []).append(e[0])
        amb = [v for v in groups.values() if len(v) > 1]
        say("paramOrder: the table's order agrees with the Torch7 dump in every tensor size (%d tensors); sizes shared by several tensors, "
            "which a dump cannot tell apart: %s" % (len(table), amb if amb else 'none'))
        return table
    pool, out = list(table), []
    for i, t in enumerate(dump):
        j = next((k for k, e in enumerate(pool) if numel(e) == int(t['numel'])), None)
        if j is None:
            raise ValueError('paramOrder: no unused tensor of %d elements for dump position %d (%s.%s)' % (t['numel'], i, t.get('type'), t.get('field')))
        out.append(pool.pop(j))
    first = next(i for i, (a, b) in enumerate(zip(out, table)) if a[0] != b[0])
    say("paramOrder: the Torch7 dump CONTRADICTS the table from position %d on (table: %s, dump: %d elements of %s) -- following the dump: %s" % (
        first, table[first][0], dump[first]['numel'], dump[first].get('type'), [e[0] for e in out]))
    return out


def flat_to_named(modelW, spec_entries, encoder=None, param_order=None, log=None):
    """Split a reference-style flat vector (getParameters(): tensors back to back, NO alignment padding) into
    this repo's named tensors, using the reference's parameter order for `encoder` (reference_order) or the override
    `param_order` (resolve_order).  encoder=None keeps this repo's own declaration order (files written with
    vdLayout = 'declaration')."""
    out, o = {}, 0
    order = resolve_order(spec_entries, encoder, param_order, log)
    total = sum(int(np.prod(shape)) for _, shape, _ in order)
    if total != len(modelW):
        raise ValueError('checkpoint holds %d parameters, the model declares %d' % (len(modelW), total))
    for name, shape, _ in order:
        n = int(np.prod(shape))
        out[name] = np.asarray(modelW[o:o + n], np.float32).reshape(shape)
        o += n
    return out


def named_to_flat(named, spec_entries, encoder=None, param_order=None):
    return np.concatenate([np.asarray(named[n], np.float32).reshape(-1)
                           for n, _, _ in resolve_order(spec_entries, encoder, param_order)])


def forget_bias_report(named, spec_entries):
    """Plausibility of a split flat vector (reported, never enforced): nn.SeqLSTM:reset() leaves the forget-gate quarter of every LSTM
    bias at 1 on a FRESH initialisation (Element-Research rnn SeqLSTM.lua, [UPSTREAM-RECALL]); a mis-ordered split of such a file puts
    other values there.  -> (n_lstm_biases, n_with_forget_quarter_all_one)"""
    n = ok = 0
    for name, shape, _ in spec_entries:
        if name.endswith('.b') and (name[:-2] + '.W') in named and len(named[name[:-2] + '.W'].shape) == 2:
            W, b = named[name[:-2] + '.W'], np.asarray(named[name]).reshape(-1)
            H4 = b.size
            if H4 % 4 == 0 and W.shape[1] == H4 and W.shape[0] > H4 // 4:       # [(D + H) x 4H] weight: an LSTM layer
                n += 1
                ok += bool(np.all(b[H4 // 4:H4 // 2] == 1.0))
    return n, ok
