"""A Torch7 `torch` global for luavm: typed, strided, 1-based tensors over numpy with Torch7's method names and conventions
(in-place methods return self, `:view` needs a contiguous tensor, `:t()` is a strided view, `:int()` on an IntTensor returns the SAME
tensor, comparison methods give ByteTensors, `:data()` is a typed `real*` cdata at the storage offset) and `torch.class`.
Only what the Lua host of this repo and a Torch7 data pipeline of the usual kind call is implemented; anything else raises.

Two deliberate strictnesses that a real Torch7 does not give you, both in the spirit of a test bed:
  * `torch.XTensor(sizes...)` is filled with a POISON pattern (NaN / 0x7f7f...), not left as whatever malloc returned;
  * when the last tensor over a storage dies, the storage is poisoned and parked (never unmapped), so a `:data()` pointer that
    outlives its tensor -- `:data()` does not pin the tensor, exactly as in Torch7 -- reads poison instead of stale-but-plausible
    values.  Python's reference counting frees at the EARLIEST moment a Lua collector could, which makes lifetime bugs deterministic.
"""
import ctypes

import numpy as np

from .interp import LuaError, LuaTable, call, first, index as lua_index_fn, tostring, type_name

DTYPES = {'Double': np.float64, 'Float': np.float32, 'Long': np.int64, 'Int': np.int32, 'Byte': np.uint8, 'Char': np.int8,
          'Short': np.int16}
CNAMES = {'Double': 'double', 'Float': 'float', 'Long': 'long', 'Int': 'int', 'Byte': 'unsigned char', 'Char': 'char', 'Short': 'short'}
_PARKED, _PARKED_BYTES = [], [0]


class Storage(object):
    __slots__ = ('base',)

    def __init__(self, base):
        self.base = base                     # the owning 1-D numpy array

    def __del__(self):
        try:
            b = self.base
            if b is None or b.size == 0:
                return
            b.view(np.uint8)[...] = 0xFF if b.dtype.kind == 'f' else 0x7F
            _PARKED.append(b)
            _PARKED_BYTES[0] += b.nbytes
            while _PARKED_BYTES[0] > (1 << 28) and len(_PARKED) > 1:
                _PARKED_BYTES[0] -= _PARKED.pop(0).nbytes
        except Exception:
            pass


def _poison(a):
    if a.size:
        a.view(np.uint8)[...] = 0xFF if a.dtype.kind == 'f' else 0x7F
    return a


class LongStorage(object):
    lua_type = 'userdata'

    def __init__(self, vals):
        self.v = [int(x) for x in vals]

    def lua_index(self, k):
        if k.__class__ is str:
            if k == 'size':
                return lambda s, *_: len(self.v)
            if k == 'totable':
                return lambda s, *_: _table(self.v)
            raise LuaError("LongStorage has no method '%s'" % k)
        i = int(k)
        if not 1 <= i <= len(self.v):
            raise LuaError('index out of bounds (LongStorage of %d, index %d)' % (len(self.v), i))
        return self.v[i - 1]

    def lua_len(self):
        return len(self.v)

    def lua_tostring(self):
        return '\n'.join(' %d' % x for x in self.v) + '\n[torch.LongStorage of size %d]\n' % len(self.v)


def _table(seq):
    t = LuaTable()
    for i, v in enumerate(seq):
        t.set(i + 1, v)
    return t


def _num(v):
    """numpy scalar -> Lua number"""
    if isinstance(v, (np.integer, int)):
        return int(v)
    f = float(v)
    return f


def _sizes(args):
    if len(args) == 1 and isinstance(args[0], LongStorage):
        return list(args[0].v)
    if len(args) == 1 and args[0].__class__ is LuaTable:
        return [int(args[0].get(i + 1)) for i in range(args[0].length())]
    out = []
    for a in args:
        if a.__class__ not in (int, float) or float(a) != int(a):
            raise LuaError('torch: a size must be an integer, got %s' % tostring(a))
        out.append(int(a))
    return out


class Tensor(object):
    lua_type = 'userdata'
    __slots__ = ('a', 'st', 'tn', 'T', '__weakref__')

    def __init__(self, T, a, tn, st=None):
        self.T, self.tn = T, tn
        if st is None:
            a = np.asarray(a)
            base = np.array(a, dtype=DTYPES[tn], copy=True, order='C').reshape(-1)      # always a fresh, owned, 1-D storage
            a = base.reshape(a.shape)
            st = Storage(base)
        self.a, self.st = a, st

    # ---- plumbing
    def new_like(self, a, tn=None):
        return Tensor(self.T, a, tn or self.tn)

    def view_of(self, a):
        return Tensor(self.T, a, self.tn, self.st)

    def typename(self):
        return 'torch.%sTensor' % self.tn

    def lua_tostring(self):
        return '%s\n[%s of size %s]\n' % (np.array2string(self.a, threshold=200), self.typename(), 'x'.join(str(s) for s in self.a.shape))

    def lua_len(self):
        return self.a.shape[0] if self.a.ndim else 0

    def lua_eq(self, a, b):
        return a is b

    def _dim(self, d, what='dimension'):
        if d.__class__ not in (int, float) or int(d) != d:
            raise LuaError('torch: bad %s %s' % (what, tostring(d)))
        d = int(d)
        if d < 0:
            d = self.a.ndim + d + 1
        if not 1 <= d <= self.a.ndim:
            raise LuaError('torch: %s %d out of range of %dD tensor' % (what, d, self.a.ndim))
        return d - 1

    def _idx(self, axis, i):
        n = self.a.shape[axis]
        if i.__class__ not in (int, float) or int(i) != i:
            raise LuaError('torch: tensor index must be an integer, got %s' % tostring(i))
        i = int(i)
        if i < 0:
            i = n + i + 1
        if not 1 <= i <= n:
            raise LuaError('index out of range: %d is outside dimension %d of size %d' % (i, axis + 1, n))
        return i - 1

    def _sub(self, key):
        """t[{...}] -> numpy view (or scalar)"""
        if key.length() > self.a.ndim:
            raise LuaError('torch: too many indices (%d) for a %dD tensor' % (key.length(), self.a.ndim))
        sl = []
        for d in range(key.length()):
            k = key.get(d + 1)
            if k.__class__ is LuaTable:
                if k.length() == 0:
                    sl.append(slice(None))
                else:
                    lo = self._idx(d, k.get(1))
                    hi = self._idx(d, k.get(2) if k.length() > 1 else k.get(1))
                    if hi < lo:
                        raise LuaError('torch: empty range in index')
                    sl.append(slice(lo, hi + 1))
            else:
                sl.append(self._idx(d, k))
        return tuple(sl)

    def lua_index(self, k):
        if k.__class__ is str:
            if k == 'THNN':
                return THNN
            if k == 'new':                              # x.new(sizes) and x:new(sizes): a tensor of x's type
                return lambda *a: TensorCtor(self.T, self.tn).lua_call([v for v in (a[1:] if a and a[0] is self else a)])
            m = getattr(Tensor, 'm_' + k, None)
            if m is None:
                raise LuaError("luavm-torch: %s has no method '%s'" % (self.typename(), k))
            return m
        if k.__class__ is LuaTable:
            r = self.a[self._sub(k)]
            return _num(r[()]) if r.ndim == 0 else self.view_of(r)
        if isinstance(k, Tensor):
            if k.tn != 'Byte':
                raise LuaError('torch: tensors can only be indexed by ByteTensor masks')
            return self.new_like(self.a[k.a.astype(bool)])
        if self.a.ndim == 0:
            raise LuaError('torch: indexing an empty tensor')
        i = self._idx(0, k)
        r = self.a[i]
        return _num(r) if self.a.ndim == 1 else self.view_of(r)

    def lua_newindex(self, k, v):
        if k.__class__ is LuaTable:
            key = self._sub(k)
        elif isinstance(k, Tensor):
            key = k.a.astype(bool)
        else:
            key = self._idx(0, k)
        if isinstance(v, Tensor):
            dst = self.a[key]
            if dst.size != v.a.size:
                raise LuaError('torch: sizes do not match in indexed assignment (%s vs %s)' % (dst.shape, v.a.shape))
            self.a[key] = v.a.reshape(dst.shape)
        elif v.__class__ in (int, float):
            self.a[key] = v
        else:
            raise LuaError('torch: cannot assign a %s into a tensor' % type_name(v))

    def lua_arith(self, op, a, b):
        if op == 'unm':
            return self.new_like(-self.a)
        x = a.a if isinstance(a, Tensor) else a
        y = b.a if isinstance(b, Tensor) else b
        ref = a if isinstance(a, Tensor) else b
        for v in (x, y):
            if not isinstance(v, np.ndarray) and v.__class__ not in (int, float):
                raise LuaError('torch: arithmetic between a tensor and a %s' % type_name(v))
        if isinstance(x, np.ndarray) and isinstance(y, np.ndarray) and x.size != y.size:
            raise LuaError('torch: inconsistent tensor size in %s' % op)
        if isinstance(x, np.ndarray) and isinstance(y, np.ndarray) and op == '*':
            if x.ndim == 2 and y.ndim == 2:
                return ref.new_like(x @ y)
            if x.ndim == 2 and y.ndim == 1:
                return ref.new_like(x @ y)
            if x.ndim == 1 and y.ndim == 1:
                return _num((x * y).sum())
        if isinstance(y, np.ndarray) and isinstance(x, np.ndarray):
            y = y.reshape(x.shape)
        if op == '+':
            r = x + y
        elif op == '-':
            r = x - y
        elif op == '*':
            r = x * y
        elif op == '/':
            if isinstance(y, np.ndarray):
                raise LuaError('torch: tensor / tensor is not defined (use cdiv)')
            r = x / y if ref.a.dtype.kind == 'f' else x // y
        else:
            raise LuaError('torch: operator %s is not defined on tensors' % op)
        return ref.new_like(r.astype(ref.a.dtype))

    # ---- shape
    def m_size(self, d=None, *_):
        if d is None:
            return LongStorage(self.a.shape)
        return int(self.a.shape[self._dim(d)])

    def m_dim(self, *_):
        return self.a.ndim
    m_nDimension = m_dim

    def m_nElement(self, *_):
        return int(self.a.size)
    m_numel = m_nElement

    def m_stride(self, d=None, *_):
        es = self.a.itemsize
        if d is None:
            return LongStorage([s // es for s in self.a.strides])
        return self.a.strides[self._dim(d)] // es

    def m_isContiguous(self, *_):
        return bool(self.a.flags.c_contiguous)

    def m_contiguous(self, *_):
        if self.a.flags.c_contiguous:
            return self
        return self.new_like(np.ascontiguousarray(self.a))

    def m_clone(self, *_):
        return self.new_like(self.a.copy())

    def m_view(self, *sizes):
        if not self.a.flags.c_contiguous:
            raise LuaError('torch: :view() expects a contiguous tensor (call :contiguous() first)')
        sz = _sizes(sizes)
        try:
            return self.view_of(self.a.reshape(sz))
        except ValueError:
            raise LuaError('torch: :view(%s) does not match %d elements' % (', '.join(str(s) for s in sz), self.a.size))

    def m_viewAs(self, other, *_):
        return Tensor.m_view(self, *other.a.shape)

    def m_reshape(self, *sizes):
        return self.new_like(np.ascontiguousarray(self.a).reshape(_sizes(sizes)).copy())

    def m_resize(self, *sizes):
        sz = _sizes(sizes)
        n = int(np.prod(sz)) if sz else 0
        if n <= self.st.base.size and self.a.flags.c_contiguous and self._offset() == 0:
            self.a = self.st.base[:n].reshape(sz)
        else:
            base = _poison(np.empty(n, DTYPES[self.tn]))
            m = min(n, self.a.size)
            base[:m] = np.ascontiguousarray(self.a).reshape(-1)[:m]
            self.st = Storage(base)
            self.a = base.reshape(sz)
        return self

    def m_resizeAs(self, other, *_):
        return Tensor.m_resize(self, *other.a.shape)

    def m_t(self, *_):
        if self.a.ndim != 2:
            raise LuaError('torch: :t() expects a 2D tensor, got %dD' % self.a.ndim)
        return self.view_of(self.a.T)

    def m_transpose(self, d1, d2, *_):
        return self.view_of(np.swapaxes(self.a, self._dim(d1), self._dim(d2)))

    def m_permute(self, *dims):
        return self.view_of(np.transpose(self.a, [self._dim(d) for d in dims]))

    def m_narrow(self, d, start, length, *_):
        ax = self._dim(d)
        lo = self._idx(ax, start)
        length = int(length)
        if length < 1 or lo + length > self.a.shape[ax]:
            raise LuaError('torch: narrow(%d, %d, %d) out of range of size %d' % (ax + 1, lo + 1, length, self.a.shape[ax]))
        sl = [slice(None)] * self.a.ndim
        sl[ax] = slice(lo, lo + length)
        return self.view_of(self.a[tuple(sl)])

    def m_sub(self, *r):
        sl = []
        for d in range(0, len(r), 2):
            lo, hi = self._idx(d // 2, r[d]), self._idx(d // 2, r[d + 1])
            sl.append(slice(lo, hi + 1))
        return self.view_of(self.a[tuple(sl)])

    def m_select(self, d, i, *_):
        ax = self._dim(d)
        r = np.take(self.a, self._idx(ax, i), axis=ax) if False else self.a[(slice(None),) * ax + (self._idx(ax, i),)]
        return _num(r) if r.ndim == 0 else self.view_of(r)

    def m_squeeze(self, d=None, *_):
        if d is None:
            return self.view_of(np.squeeze(self.a))
        ax = self._dim(d)
        return self.view_of(np.squeeze(self.a, ax)) if self.a.shape[ax] == 1 else self

    def m_expandAs(self, other, *_):
        return self.view_of(np.broadcast_to(self.a, other.a.shape))

    def m_expand(self, *sizes):
        return self.view_of(np.broadcast_to(self.a, _sizes(sizes)))

    def m_repeatTensor(self, *reps):
        return self.new_like(np.tile(self.a, _sizes(reps)))

    def _offset(self):
        return (self.a.__array_interface__['data'][0] - self.st.base.__array_interface__['data'][0]) // self.a.itemsize

    def m_storageOffset(self, *_):
        return self._offset() + 1

    def m_data(self, *_):
        if self.a.size == 0:
            return self.T.vm.ffi.pointer_to(CNAMES[self.tn], 0)
        return self.T.vm.ffi.pointer_to(CNAMES[self.tn], self.a.__array_interface__['data'][0])       # NOT pinned

    # ---- types
    def _as(self, tn):
        if tn == self.tn:
            return self
        return self.new_like(self.a.astype(DTYPES[tn]), tn)

    def m_double(self, *_):
        return self._as('Double')

    def m_float(self, *_):
        return self._as('Float')

    def m_long(self, *_):
        return self._as('Long')

    def m_int(self, *_):
        return self._as('Int')

    def m_byte(self, *_):
        return self._as('Byte')

    def m_short(self, *_):
        return self._as('Short')

    def m_char(self, *_):
        return self._as('Char')

    def m_type(self, tn=None, *_):
        if tn is None:
            return self.typename()
        return self._as(tn.replace('torch.', '').replace('Tensor', ''))

    def m_typeAs(self, other, *_):
        return self._as(other.tn)

    def m_cuda(self, *_):
        raise LuaError('luavm-torch: there is no cutorch here (:cuda() called)')

    # ---- fill / copy
    def m_fill(self, v, *_):
        self.a[...] = v
        return self

    def m_zero(self, *_):
        self.a[...] = 0
        return self

    def m_copy(self, src, *_):
        if not isinstance(src, Tensor):
            raise LuaError('torch: :copy() expects a tensor, got %s' % type_name(src))
        if src.a.size != self.a.size:
            raise LuaError('torch: :copy() sizes do not match (%s <- %s)' % (self.a.shape, src.a.shape))
        self.a[...] = np.ascontiguousarray(src.a).reshape(self.a.shape).astype(self.a.dtype)
        return self

    def m_set(self, src, *_):
        self.a, self.st = src.a, src.st
        return self

    def m_random(self, lo=None, hi=None, *_):
        rs = self.T.rng
        if lo is None:
            lo, hi = 1, 2 ** 31 - 1 if self.a.dtype.kind != 'u' else 255
        elif hi is None:
            lo, hi = 1, lo
        self.a[...] = rs.randint(int(lo), int(hi) + 1, size=self.a.shape)
        return self

    def m_uniform(self, lo=0, hi=1, *_):
        self.a[...] = self.T.rng.uniform(lo, hi, size=self.a.shape)
        return self

    def m_normal(self, mean=0, std=1, *_):
        self.a[...] = self.T.rng.normal(mean, std, size=self.a.shape)
        return self

    # ---- comparisons (-> ByteTensor)
    def _cmp(self, other, fn):
        y = other.a.reshape(self.a.shape) if isinstance(other, Tensor) else other
        return self.new_like(fn(self.a, y).astype(np.uint8), 'Byte')

    def m_gt(self, o, *_):
        return self._cmp(o, np.greater)

    def m_lt(self, o, *_):
        return self._cmp(o, np.less)

    def m_ge(self, o, *_):
        return self._cmp(o, np.greater_equal)

    def m_le(self, o, *_):
        return self._cmp(o, np.less_equal)

    def m_eq(self, o, *_):
        return self._cmp(o, np.equal)

    def m_ne(self, o, *_):
        return self._cmp(o, np.not_equal)

    # ---- reductions
    def _reduce(self, fn, d, with_index=None):
        if d is None:
            return _num(fn(self.a))
        ax = self._dim(d)
        vals = self.new_like(np.expand_dims(fn(self.a, axis=ax), ax))
        if with_index is None:
            return vals
        idx = self.new_like(np.expand_dims(with_index(self.a, axis=ax) + 1, ax), 'Long')
        return (vals, idx)

    def m_sum(self, d=None, *_):
        if d is None:
            return _num(self.a.sum(dtype=np.float64) if self.a.dtype.kind == 'f' else self.a.sum(dtype=np.int64))
        return self._reduce(np.sum, d)

    def m_mean(self, d=None, *_):
        if d is None:
            return float(self.a.mean(dtype=np.float64))
        ax = self._dim(d)
        return self.new_like(np.expand_dims(self.a.mean(axis=ax), ax).astype(self.a.dtype))

    def m_max(self, d=None, *_):
        return self._reduce(np.max, d, np.argmax)

    def m_min(self, d=None, *_):
        return self._reduce(np.min, d, np.argmin)

    def m_median(self, d=None, *_):
        # Torch7: the LOWER median (element (n-1)/2 of the sorted values), along the last dimension by default
        ax = self.a.ndim - 1 if d is None else self._dim(d)
        order = np.argsort(self.a, axis=ax, kind='stable')
        k = (self.a.shape[ax] - 1) // 2
        idx = np.take(order, [k], axis=ax)
        vals = np.take_along_axis(self.a, idx, axis=ax)
        return (self.new_like(vals), self.new_like(idx + 1, 'Long'))

    def m_sort(self, d=None, descending=None, *_):
        if d.__class__ is bool:
            d, descending = None, d
        ax = self.a.ndim - 1 if d is None else self._dim(d)
        if descending:
            order = np.argsort(-self.a.astype(np.float64), axis=ax, kind='stable')
        else:
            order = np.argsort(self.a, axis=ax, kind='stable')
        return (self.new_like(np.take_along_axis(self.a, order, axis=ax)), self.new_like(order + 1, 'Long'))

    def m_topk(self, k=1, *args):
        """torch.topk(x, k [, dim] [, dir] [, sort]): dir = true -> the k LARGEST.  Torch7 leaves their order unspecified unless sort is
        given; here they come sorted (descending for dir = true), which is one of the orders Torch7 may produce."""
        dim = [a for a in args if a.__class__ is not bool and a is not None]
        flags = [a for a in args if a.__class__ is bool]
        largest = bool(flags[0]) if flags else False
        ax = self.a.ndim - 1 if not dim else self._dim(dim[0])
        key = -self.a.astype(np.float64) if largest else self.a.astype(np.float64)
        order = np.take(np.argsort(key, axis=ax, kind='stable'), np.arange(int(k)), axis=ax)
        return (self.new_like(np.take_along_axis(self.a, order, axis=ax)), self.new_like(order + 1, 'Long'))

    def m_norm(self, p=2, *_):
        return float(np.linalg.norm(self.a.reshape(-1).astype(np.float64), p))

    def m_std(self, *_):
        return float(self.a.std(ddof=1))

    def m_all(self, *_):
        return bool(self.a.all())

    def m_any(self, *_):
        return bool(self.a.any())

    # ---- in-place arithmetic
    def _other(self, o):
        if isinstance(o, Tensor):
            if o.a.size != self.a.size:
                raise LuaError('torch: inconsistent tensor sizes (%s vs %s)' % (self.a.shape, o.a.shape))
            return o.a.reshape(self.a.shape)
        if o.__class__ in (int, float):
            return o
        raise LuaError('torch: number or tensor expected, got %s' % type_name(o))

    def _inplace(self, r):
        self.a[...] = r.astype(self.a.dtype) if isinstance(r, np.ndarray) else r
        return self

    def m_add(self, x, y=None, z=None, *_):
        if z is not None:                                        # add(tensor1, value, tensor2): self = t1 + v * t2
            return self._inplace(x.a + y * z.a.reshape(x.a.shape))
        if y is None:
            return self._inplace(self.a + self._other(x))
        if isinstance(x, Tensor):                                # add(tensor1, tensor2) or add(tensor1, value)
            return self._inplace(x.a + (y.a.reshape(x.a.shape) if isinstance(y, Tensor) else y))
        return self._inplace(self.a + x * self._other(y))        # add(value, tensor)

    def m_csub(self, x, *_):
        return self._inplace(self.a - self._other(x))

    def m_mul(self, x, y=None, *_):
        if y is not None:
            return self._inplace(x.a * y)
        return self._inplace(self.a * x)

    def m_div(self, x, y=None, *_):
        if y is not None:
            return self._inplace(x.a / y)
        return self._inplace(self.a / x if self.a.dtype.kind == 'f' else self.a // x)

    def m_cmul(self, x, y=None, *_):
        if y is not None:
            return self._inplace(x.a * y.a.reshape(x.a.shape))
        return self._inplace(self.a * self._other(x))

    def m_cdiv(self, x, y=None, *_):
        if y is not None:
            return self._inplace(x.a / y.a.reshape(x.a.shape))
        return self._inplace(self.a / self._other(x))

    def m_cinv(self, *_):
        return self._inplace(1.0 / self.a)

    def m_sqrt(self, *_):
        return self._inplace(np.sqrt(self.a))

    def m_pow(self, p, *_):
        return self._inplace(np.power(self.a, p))

    def m_exp(self, *_):
        return self._inplace(np.exp(self.a))

    def m_log(self, *_):
        return self._inplace(np.log(self.a))

    def m_abs(self, *_):
        return self._inplace(np.abs(self.a))

    def m_neg(self, *_):
        return self._inplace(-self.a)

    def m_clamp(self, lo, hi, *_):
        return self._inplace(np.clip(self.a, lo, hi))

    def m_cumsum(self, d=1, *_):
        return self.new_like(np.cumsum(self.a, axis=self._dim(d)))

    def m_dot(self, o, *_):
        return _num((self.a.reshape(-1).astype(np.float64) * o.a.reshape(-1)).sum())

    # ---- gather / index
    def m_index(self, d, idx, *_):
        if not isinstance(idx, Tensor) or idx.tn != 'Long':
            raise LuaError('torch: :index() expects a LongTensor of indices, got %s' % (idx.typename() if isinstance(idx, Tensor) else type_name(idx)))
        ax = self._dim(d)
        ii = idx.a.reshape(-1)
        if ii.size and (ii.min() < 1 or ii.max() > self.a.shape[ax]):
            raise LuaError('torch: :index() index out of range (1..%d)' % self.a.shape[ax])
        return self.new_like(np.take(self.a, ii - 1, axis=ax))

    def m_indexCopy(self, d, idx, src, *_):
        ax = self._dim(d)
        sl = [slice(None)] * self.a.ndim
        sl[ax] = idx.a.reshape(-1) - 1
        self.a[tuple(sl)] = src.a
        return self

    def m_gather(self, d, idx, *_):
        ax = self._dim(d)
        if idx.tn != 'Long':
            raise LuaError('torch: :gather() expects a LongTensor index')
        if idx.a.size and (idx.a.min() < 1 or idx.a.max() > self.a.shape[ax]):
            raise LuaError('torch: :gather() index out of range')
        return self.new_like(np.take_along_axis(self.a, idx.a - 1, axis=ax))

    def m_scatter(self, d, idx, src, *_):
        ax = self._dim(d)
        np.put_along_axis(self.a, idx.a - 1, src.a if isinstance(src, Tensor) else src, axis=ax)
        return self

    def m_maskedFill(self, mask, v, *_):
        self.a[mask.a.reshape(self.a.shape).astype(bool)] = v
        return self

    def m_maskedSelect(self, mask, *_):
        return self.new_like(self.a[mask.a.reshape(self.a.shape).astype(bool)])

    def m_nonzero(self, *_):
        return self.new_like(np.argwhere(self.a) + 1, 'Long')

    def m_totable(self, *_):
        return self.T.totable(self)

    def m_cdata(self, *_):
        return self                                  # (the THNN stub below takes the tensors themselves)

    def m_new(self, *sizes):
        return TensorCtor(self.T, self.tn).lua_call(list(sizes))

    def m_addcmul(self, v, a, b=None, *_):
        if b is None:
            v, a, b = 1, v, a
        return self._inplace(self.a + v * a.a.reshape(self.a.shape) * b.a.reshape(self.a.shape))

    def m_addcdiv(self, v, a, b=None, *_):
        if b is None:
            v, a, b = 1, v, a
        return self._inplace(self.a + v * a.a.reshape(self.a.shape) / b.a.reshape(self.a.shape))

    def m_isSameSizeAs(self, o, *_):
        return self.a.shape == o.a.shape

    def m_storage(self, *_):
        return self.st

    def m_apply(self, fn, *_):
        flat = self.a.reshape(-1) if self.a.flags.c_contiguous else None
        it = np.nditer(self.a, op_flags=['readwrite'])
        for x in it:
            r = first(call(fn, [_num(x[()])]))
            if r is not None:
                x[...] = r
        del flat
        return self


class _THNN(object):
    """`tensor.THNN.<fn>(tensor:cdata(), ...)`: the two THNN entry points model_utils/MaskSoftMax.lua calls through the FFI"""
    lua_type = 'table'

    def lua_index(self, k):
        return getattr(self, k, None)

    @staticmethod
    def SoftMax_updateOutput(inp, out, *_):
        x = inp.a.astype(np.float64)
        e = np.exp(x - x.max(-1, keepdims=True))
        Tensor.m_resizeAs(out, inp)
        out.a[...] = e / e.sum(-1, keepdims=True)

    @staticmethod
    def SoftMax_updateGradInput(inp, gout, gin, out, *_):
        y, g = out.a.astype(np.float64), gout.a.astype(np.float64)
        Tensor.m_resizeAs(gin, out)
        gin.a[...] = y * (g - (g * y).sum(-1, keepdims=True))


THNN = _THNN()


class TensorCtor(object):
    """torch.FloatTensor & co: callable, and a method table (torch.FloatTensor.new ... are not needed here)"""
    lua_type = 'table'

    def __init__(self, T, tn):
        self.T, self.tn = T, tn

    def lua_tostring(self):
        return 'torch.%sTensor' % self.tn

    def lua_index(self, k):
        m = getattr(Tensor, 'm_' + k, None) if k.__class__ is str else None
        if k == 'new':
            return lambda *a: self.lua_call(list(a))
        return m

    def lua_call(self, args):
        T, tn = self.T, self.tn
        if not args:
            return Tensor(T, np.zeros(0, DTYPES[tn]), tn)
        a0 = args[0]
        if isinstance(a0, Tensor):
            if a0.tn != tn:
                raise LuaError('torch.%sTensor(tensor): expected a %sTensor, got %s' % (tn, tn, a0.typename()))
            return a0.view_of(a0.a)                              # shares the storage
        if a0.__class__ is LuaTable:
            arr = np.array(T.from_table(a0), dtype=DTYPES[tn])
            return Tensor(T, arr, tn)
        sz = _sizes(args)
        return Tensor(T, _poison(np.empty(sz, DTYPES[tn])), tn)


class Torch(object):
    def __init__(self, vm, seed=0):
        self.vm = vm
        self.rng = np.random.RandomState(seed)
        self.default = 'Double'
        self.classes = {}
        t = self.module = LuaTable()
        for tn in DTYPES:
            t.set(tn + 'Tensor', TensorCtor(self, tn))
        t.set('LongStorage', lambda *a: LongStorage(_sizes(a) if not (len(a) == 1 and a[0].__class__ in (int, float)) else [0] * int(a[0])))
        for name in ('Tensor', 'zeros', 'ones', 'range', 'repeatTensor', 'totable', 'multinomial', 'cat', 'sum', 'mean', 'max', 'min',
                     'sqrt', 'median', 'cmul', 'cdiv', 'le', 'lt', 'ge', 'gt', 'eq', 'ne', 'manualSeed', 'setdefaulttensortype', 'CmdLine',
                     'getdefaulttensortype', 'type', 'typename', 'isTensor', 'class', 'random', 'randperm', 'rand', 'randn', 'uniform',
                     'sort', 'topk', 'abs', 'exp', 'log', 'add', 'mul', 'div', 'dot', 'norm', 'cumsum', 'setnumthreads', 'getnumthreads',
                     'save', 'load', 'isTypeOf', 'setmetatable', 'getmetatable', 'squeeze', 'floor', 'clamp', 'pow', 'seed', 'triu', 'tril',
                     'pointer'):
            t.set(name, getattr(self, 'f_' + name.replace('class', 'class_')))
        vm.globals.set('torch', t)
        vm.loaded.set('torch', t)
        vm.torch = self

    # ---- helpers
    def tensor(self, arr, tn=None):
        """numpy array -> Tensor (copy); tn defaults from the dtype"""
        arr = np.asarray(arr)
        if tn is None:
            tn = {v: k for k, v in DTYPES.items()}.get(arr.dtype.type)
            if tn is None:
                raise LuaError('luavm-torch: no tensor type for dtype %s' % arr.dtype)
        return Tensor(self, np.array(arr, dtype=DTYPES[tn], copy=True), tn)

    def from_table(self, t):
        if t.__class__ is LuaTable:
            return [self.from_table(t.get(i + 1)) for i in range(t.length())]
        if t.__class__ in (int, float):
            return t
        raise LuaError('torch: a tensor can only be built from (nested) tables of numbers, found %s' % type_name(t))

    def totable(self, x):
        def rec(a):
            if a.ndim == 1:
                return _table([_num(v) for v in a])
            return _table([rec(s) for s in a])
        if x.a.ndim == 0:
            return _table([])
        return rec(x.a)

    # ---- torch.* functions
    def f_Tensor(self, *args):
        return TensorCtor(self, self.default).lua_call(list(args))

    def f_zeros(self, *sizes):
        return Tensor(self, np.zeros(_sizes(sizes), DTYPES[self.default]), self.default)

    def f_ones(self, *sizes):
        return Tensor(self, np.ones(_sizes(sizes), DTYPES[self.default]), self.default)

    def f_range(self, a, b, step=1, *_):
        n = int((b - a) // step) + 1
        return Tensor(self, np.asarray([a + i * step for i in range(max(n, 0))], DTYPES[self.default]), self.default)

    def f_repeatTensor(self, t, *reps):
        return Tensor.m_repeatTensor(t, *reps)

    def f_totable(self, t, *_):
        return self.totable(t)

    def f_multinomial(self, probs, n, replacement=False, *_):
        p = probs.a.astype(np.float64)
        p = p / p.sum(-1, keepdims=True)
        if p.ndim == 1:
            r = self.rng.choice(p.size, size=int(n), replace=bool(replacement), p=p) + 1
        else:
            r = np.stack([self.rng.choice(p.shape[1], size=int(n), replace=bool(replacement), p=row) + 1 for row in p])
        return Tensor(self, r.astype(np.int64), 'Long')

    def f_cat(self, a, b=None, d=None, *_):
        if a.__class__ is LuaTable:
            ts = [a.get(i + 1) for i in range(a.length())]
            d = b
        else:
            ts = [a, b]
        ax = (ts[0].a.ndim if d is None else int(d)) - 1
        return ts[0].new_like(np.concatenate([t.a for t in ts], axis=ax))

    def _fn(name):
        def f(self, t, *args):
            if not isinstance(t, Tensor):
                raise LuaError('torch.%s: tensor expected, got %s' % (name, type_name(t)))
            return getattr(Tensor, 'm_' + name)(t.new_like(t.a.copy()) if name in Torch._COPYING else t, *args)
        return f
    _COPYING = ('sqrt', 'abs', 'exp', 'log', 'cumsum', 'clamp', 'pow')
    for _n in ('sum', 'mean', 'max', 'min', 'median', 'le', 'lt', 'ge', 'gt', 'eq', 'ne', 'sort', 'topk', 'sqrt', 'abs', 'exp', 'log', 'dot',
               'norm', 'cumsum', 'squeeze', 'clamp', 'pow'):
        locals()['f_' + _n] = _fn(_n)
    del _fn, _n

    def f_cmul(self, a, b, *_):
        return a.new_like(a.a * b.a.reshape(a.a.shape))

    def f_cdiv(self, a, b, *_):
        return a.new_like(a.a / b.a.reshape(a.a.shape))

    def f_add(self, a, b, *_):
        return a.lua_arith('+', a, b)

    def f_mul(self, a, b, *_):
        return a.new_like(a.a * b)

    def f_div(self, a, b, *_):
        return a.new_like(a.a / b)

    def f_floor(self, a, *_):
        return a.new_like(np.floor(a.a))

    def f_triu(self, a, k=0, *_):
        return a.new_like(np.triu(a.a, int(k)))

    def f_tril(self, a, k=0, *_):
        return a.new_like(np.tril(a.a, int(k)))

    def f_pointer(self, o, *_):
        return id(o)

    def f_manualSeed(self, s=0, *_):
        self.rng = np.random.RandomState(int(s) % (2 ** 32))

    def f_seed(self, *_):
        return 0

    def f_setdefaulttensortype(self, name, *_):
        self.default = name.replace('torch.', '').replace('Tensor', '')

    def f_getdefaulttensortype(self, *_):
        return 'torch.%sTensor' % self.default

    def f_setnumthreads(self, *_):
        return None

    def f_getnumthreads(self, *_):
        return 1

    def f_random(self, a=None, b=None, *_):
        if a is None:
            return int(self.rng.randint(1, 2 ** 31))
        if b is None:
            return int(self.rng.randint(1, int(a) + 1))
        return int(self.rng.randint(int(a), int(b) + 1))

    def f_uniform(self, a=0, b=1, *_):
        return float(self.rng.uniform(a, b))

    def f_randperm(self, n, *_):
        return Tensor(self, (self.rng.permutation(int(n)) + 1).astype(DTYPES[self.default]), self.default)

    def f_rand(self, *sizes):
        return Tensor(self, self.rng.uniform(size=_sizes(sizes)).astype(DTYPES[self.default]), self.default)

    def f_randn(self, *sizes):
        return Tensor(self, self.rng.normal(size=_sizes(sizes)).astype(DTYPES[self.default]), self.default)

    def f_type(self, v=None, *_):
        tn = self.f_typename(v)
        return tn if tn is not None else type_name(v)

    def f_typename(self, v=None, *_):
        if isinstance(v, Tensor):
            return v.typename()
        if isinstance(v, LongStorage):
            return 'torch.LongStorage'
        if v.__class__ is LuaTable and v.meta is not None:
            n = lua_index_fn(v.meta, '__typename')
            return n
        return None

    def f_isTensor(self, v=None, *_):
        return isinstance(v, Tensor)

    def f_isTypeOf(self, v, name, *_):
        if isinstance(v, Tensor):
            return v.typename() == name
        m = v.meta if v.__class__ is LuaTable else None
        while m is not None:
            if m.get('__typename') == name:
                return True
            m = m.meta.get('__index') if m.meta is not None else None          # the parent class (f_class_)
        return False

    def f_setmetatable(self, t, name, *_):
        t.meta = self.classes[name]
        return t

    def f_getmetatable(self, name, *_):
        return self.classes.get(name)

    def f_class_(self, name=None, parent=None, *_):
        """torch.class(name [, parentName]) -> metatable [, parent metatable]; the constructor `name(...)` goes into the global
        namespace (dotted names into their table), instances are tables with this metatable, `__init` is the initialiser"""
        if name.__class__ is not str:
            raise LuaError("bad argument #1 to 'torch.class' (string expected)")
        mt = LuaTable()
        mt.set('__index', mt)
        mt.set('__typename', name)
        pmt = None
        if parent is not None:
            pmt = self.classes.get(parent)
            if pmt is None:
                raise LuaError("torch.class: parent class '%s' is not defined" % parent)
            holder = LuaTable()
            holder.set('__index', pmt)
            mt.meta = holder
        self.classes[name] = mt

        def forward_call(obj, *args):
            f = lua_index_fn(obj, '__call__')
            if f is None:
                raise LuaError("attempt to call an instance of class '%s'" % name)
            return call(f, [obj] + list(args))
        mt.set('__call', forward_call)
        ctor = LuaTable()
        cm = LuaTable()

        def construct(_self, *args):
            obj = LuaTable()
            obj.meta = mt
            init = lua_index_fn(obj, '__init')
            if init is not None:
                call(init, [obj] + list(args))
            return obj
        cm.set('__call', construct)
        cm.set('__index', mt)
        cm.set('__newindex', mt)
        ctor.meta = cm
        mt.set('new', lambda *args: construct(None, *args))
        parts = name.split('.')
        scope = self.vm.globals
        for part in parts[:-1]:
            nxt = scope.get(part)
            if nxt is None:
                nxt = LuaTable()
                scope.set(part, nxt)
            scope = nxt
        scope.set(parts[-1], ctor)
        return (mt, pmt) if pmt is not None else mt

    def f_CmdLine(self, *_):
        """torch.CmdLine(): text / option / parse, as opts.lua uses them (values typed like the option's default)"""
        T = self

        class CmdLine(object):
            lua_type = 'table'

            def __init__(self):
                self.opts = []

            def lua_index(self, k):
                if k == 'text':
                    return lambda *_a: None
                if k == 'option':
                    return lambda _s, name, default=None, help='', *_a: self.opts.append((name, default))
                if k == 'parse':
                    def parse(_s, arg=None, *_a):
                        out = LuaTable()
                        for name, default in self.opts:
                            out.set(name.lstrip('-'), default)
                        args = []
                        i = 1
                        while arg is not None and arg.get(i) is not None:
                            args.append(arg.get(i))
                            i += 1
                        known = dict(self.opts)
                        i = 0
                        while i < len(args):
                            if args[i] not in known:
                                raise LuaError('CmdLine: unknown option %s' % (args[i],))
                            d, v = known[args[i]], args[i + 1]
                            if d.__class__ is bool:
                                v = v in ('true', True)
                            elif isinstance(d, (int, float)):
                                v = float(v)
                                v = int(v) if v == int(v) else v
                            out.set(args[i].lstrip('-'), v)
                            i += 2
                        return out
                    return parse
                return None
        return CmdLine()

    def f_save(self, *_):
        raise LuaError('luavm-torch: torch.save is not available (install a handler on vm.torch.module)')

    def f_load(self, *_):
        raise LuaError('luavm-torch: torch.load is not available (install a handler on vm.torch.module)')


def install(vm, seed=0):
    return Torch(vm, seed)
