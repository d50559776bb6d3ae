"""LuaJIT's `ffi` library for luavm, over ctypes: ffi.cdef / load / new / cast / typeof / sizeof / string / gc / copy / fill /
istype / C, cdata objects (pointers, arrays, structs, boxed 64-bit integers, C functions) and LuaJIT's conversion rules
(doc/ext_ffi_semantics: Lua -> C on assignment / call, C -> Lua on read / return, pointer compatibility incl. discarded
qualifiers, string -> const char*, nil -> NULL, NULL == nil, 64-bit integers boxed, single-initialiser replication for arrays).
Deliberately STRICTER than LuaJIT in one place: indexing an array cdata whose length is known is bounds-checked (LuaJIT reads
whatever is there) -- an out-of-range index in the Lua host is a bug this VM should report, not execute.
"""
import ctypes
import re
import weakref

from .interp import LuaError, LuaTable, call, fmt_number, type_name

# ---------------------------------------------------------------------------------------------------------------------- C types


class CT(object):
    const = False
    size = None
    align = 1

    def with_const(self):
        import copy
        c = copy.copy(self)
        c.const = True
        return c


class Void(CT):
    size = 1                                     # GNU C: sizeof(void) == 1 (pointer arithmetic on void* moves bytes)

    def __str__(self):
        return ('const ' if self.const else '') + 'void'


class Num(CT):
    def __init__(self, name, ctype, kind, signed=True):
        self.name, self.ctype, self.kind, self.signed = name, ctype, kind, signed      # kind: 'int' | 'float' | 'bool'
        self.size = self.align = ctypes.sizeof(ctype)

    def __str__(self):
        return ('const ' if self.const else '') + self.name


class Ptr(CT):
    size = align = 8

    def __init__(self, to):
        self.to = to

    def __str__(self):
        return '%s *%s' % (self.to, ' const' if self.const else '')


class Arr(CT):
    def __init__(self, of, n):
        self.of, self.n = of, n                  # n None = variable length ('?') until instantiated
        self.align = of.align
        self.size = None if n is None else of.size * n

    def __str__(self):
        return '%s [%s]' % (self.of, '?' if self.n is None else self.n)


class Struct(CT):
    def __init__(self, name):
        self.name, self.fields, self.index = name, None, {}          # fields None = incomplete (opaque) type

    def define(self, fields):
        off, align, out = 0, 1, []
        for fname, ct in fields:
            if ct.size is None:
                raise LuaError('ffi: field %s of struct %s has incomplete type' % (fname, self.name))
            off = (off + ct.align - 1) // ct.align * ct.align
            out.append((fname, ct, off))
            self.index[fname] = (ct, off)
            off += ct.size
            align = max(align, ct.align)
        self.fields, self.align = out, align
        self.size = (off + align - 1) // align * align

    def __str__(self):
        return ('const ' if self.const else '') + 'struct ' + self.name


class Func(CT):
    def __init__(self, ret, params, vararg=False):
        self.ret, self.params, self.vararg = ret, params, vararg

    def __str__(self):
        return '%s (*)(%s)' % (self.ret, ', '.join(str(p) for p in self.params))


def _num(name, ct, kind, signed=True):
    return Num(name, ct, kind, signed)


BASE_TYPES = {
    'char': _num('char', ctypes.c_int8, 'int'), 'signed char': _num('signed char', ctypes.c_int8, 'int'),
    'unsigned char': _num('unsigned char', ctypes.c_uint8, 'int', False),
    'short': _num('short', ctypes.c_int16, 'int'), 'unsigned short': _num('unsigned short', ctypes.c_uint16, 'int', False),
    'int': _num('int', ctypes.c_int32, 'int'), 'unsigned int': _num('unsigned int', ctypes.c_uint32, 'int', False),
    'long': _num('long', ctypes.c_int64, 'int'), 'unsigned long': _num('unsigned long', ctypes.c_uint64, 'int', False),
    'long long': _num('long long', ctypes.c_int64, 'int'), 'unsigned long long': _num('unsigned long long', ctypes.c_uint64, 'int', False),
    'float': _num('float', ctypes.c_float, 'float'), 'double': _num('double', ctypes.c_double, 'float'),
    'bool': _num('bool', ctypes.c_bool, 'bool', False), '_Bool': _num('bool', ctypes.c_bool, 'bool', False),
    'int8_t': _num('int8_t', ctypes.c_int8, 'int'), 'uint8_t': _num('uint8_t', ctypes.c_uint8, 'int', False),
    'int16_t': _num('int16_t', ctypes.c_int16, 'int'), 'uint16_t': _num('uint16_t', ctypes.c_uint16, 'int', False),
    'int32_t': _num('int32_t', ctypes.c_int32, 'int'), 'uint32_t': _num('uint32_t', ctypes.c_uint32, 'int', False),
    'int64_t': _num('int64_t', ctypes.c_int64, 'int'), 'uint64_t': _num('uint64_t', ctypes.c_uint64, 'int', False),
    'size_t': _num('size_t', ctypes.c_uint64, 'int', False), 'ssize_t': _num('ssize_t', ctypes.c_int64, 'int'),
    'intptr_t': _num('intptr_t', ctypes.c_int64, 'int'), 'uintptr_t': _num('uintptr_t', ctypes.c_uint64, 'int', False),
    'ptrdiff_t': _num('ptrdiff_t', ctypes.c_int64, 'int'),
}
_TOK = re.compile(r'\s*(?:(\.\.\.)|([A-Za-z_][A-Za-z_0-9]*)|(0[xX][0-9a-fA-F]+|\d+)|(.))', re.S)


def _ctokens(src):
    src = re.sub(r'/\*.*?\*/', ' ', src, flags=re.S)
    src = re.sub(r'//[^\n]*', ' ', src)
    out = []
    for m in _TOK.finditer(src):
        if m.group(1):
            out.append(('op', '...'))
        elif m.group(2):
            out.append(('id', m.group(2)))
        elif m.group(3):
            out.append(('num', int(m.group(3), 0)))
        elif m.group(4) and not m.group(4).isspace():
            out.append(('op', m.group(4)))
    out.append(('eof', ''))
    return out


class CParser(object):
    """declarations of the kind a LuaJIT cdef holds: typedefs, struct definitions, function prototypes, static const ints"""

    def __init__(self, ffi, src):
        self.ffi, self.t, self.p = ffi, _ctokens(src), 0

    def peek(self):
        return self.t[self.p]

    def accept(self, text):
        if self.t[self.p][1] == text and self.t[self.p][0] != 'num':
            self.p += 1
            return True
        return False

    def expect(self, text):
        if not self.accept(text):
            raise LuaError("ffi: '%s' expected near '%s'" % (text, self.peek()[1]))

    def specifiers(self):
        """-> (base CT, is_typedef, is_static)"""
        is_typedef = is_static = const = False
        words, base = [], None
        while True:
            kind, tx = self.peek()
            if kind != 'id':
                break
            if tx == 'typedef':
                is_typedef = True
            elif tx in ('static', 'extern', 'inline', 'volatile', 'restrict', '__restrict', 'register'):
                is_static = is_static or tx == 'static'
            elif tx == 'const':
                const = True
            elif tx in ('struct', 'union'):
                self.p += 1
                base = self.struct_spec()
                continue
            elif tx in ('unsigned', 'signed', 'short', 'long', 'int', 'char', 'float', 'double', 'void', 'bool', '_Bool') and base is None:
                words.append(tx)
            elif base is None and not words and tx in self.ffi.typedefs:
                base = self.ffi.typedefs[tx]
            elif base is None and not words and tx in BASE_TYPES:
                base = BASE_TYPES[tx]
            else:
                break                                           # the declarator's name
            self.p += 1
        if base is None:
            if not words:
                raise LuaError("ffi: declaration specifier expected near '%s'" % self.peek()[1])
            if words == ['void']:
                base = Void()
            else:
                w = [x for x in words if x not in ('int', 'signed')] if words not in (['int'], ['signed'], ['signed', 'int']) else ['int']
                if 'char' in words and 'signed' in words:
                    w = ['signed', 'char']
                key = ' '.join(sorted(w, key=lambda x: {'unsigned': 0, 'signed': 0}.get(x, 1)))
                if key == 'unsigned':
                    key = 'unsigned int'
                if key not in BASE_TYPES:
                    raise LuaError('ffi: unknown type %r' % ' '.join(words))
                base = BASE_TYPES[key]
        if const:
            base = base.with_const()
        return base, is_typedef, is_static

    def struct_spec(self):
        kind, tx = self.peek()
        name = None
        if kind == 'id':
            name = tx
            self.p += 1
        st = self.ffi.structs.get(name) if name else None
        if st is None:
            st = Struct(name or '<anonymous>')
            if name:
                self.ffi.structs[name] = st
        if self.accept('{'):
            fields = []
            while not self.accept('}'):
                base, _, _ = self.specifiers()
                while True:
                    ct, fname = self.declarator(base)
                    fields.append((fname, ct))
                    if not self.accept(','):
                        break
                self.expect(';')
            st.define(fields)
        return st

    def declarator(self, base):
        """-> (CT, name or None)"""
        ct = base
        while self.accept('*'):
            ct = Ptr(ct)
            while self.peek()[1] in ('const', 'volatile', 'restrict', '__restrict') and self.peek()[0] == 'id':
                if self.peek()[1] == 'const':
                    ct = ct.with_const()
                self.p += 1
        name = None
        inner = None
        if self.peek() == ('op', '(') and self.t[self.p + 1] == ('op', '*'):       # (*name)(params): pointer to function
            self.p += 2
            if self.peek()[0] == 'id':
                name = self.peek()[1]
                self.p += 1
            self.expect(')')
            inner = 'fptr'
        elif self.peek()[0] == 'id':
            name = self.peek()[1]
            self.p += 1
        if self.accept('('):
            params, vararg = [], False
            if not self.accept(')'):
                while True:
                    if self.accept('...'):
                        vararg = True
                        break
                    pbase, _, _ = self.specifiers()
                    pct, _ = self.declarator(pbase)
                    if isinstance(pct, Arr):
                        pct = Ptr(pct.of)                        # array parameters decay
                    if not (isinstance(pct, Void) and self.peek()[1] == ')' and not params):
                        params.append(pct)
                    if not self.accept(','):
                        break
                self.expect(')')
            ct = Func(ct, params, vararg)
            if inner == 'fptr':
                ct = Ptr(ct)
            return ct, name
        dims = []
        while self.accept('['):
            if self.accept('?'):
                dims.append(None)
            elif self.peek()[0] == 'num':
                dims.append(self.peek()[1])
                self.p += 1
            else:
                dims.append(None)
            self.expect(']')
        for n in reversed(dims):
            ct = Arr(ct, n)
        return ct, name

    def declarations(self):
        while self.peek()[0] != 'eof':
            if self.accept(';'):
                continue
            base, is_typedef, is_static = self.specifiers()
            if self.accept(';'):
                continue                                         # `struct x {...};`
            while True:
                ct, name = self.declarator(base)
                if name is None:
                    raise LuaError("ffi: declarator name expected near '%s'" % self.peek()[1])
                if is_typedef:
                    self.ffi.typedefs[name] = ct
                elif self.accept('='):
                    sign = -1 if self.accept('-') else 1
                    kind, v = self.peek()
                    if kind != 'num':
                        raise LuaError('ffi: constant expected for %s' % name)
                    self.p += 1
                    self.ffi.constants[name] = sign * v
                else:
                    self.ffi.decls[name] = ct
                if not self.accept(','):
                    break
            self.expect(';')

    def abstract_type(self):
        base, _, _ = self.specifiers()
        ct, name = self.declarator(base)
        if self.peek()[0] != 'eof':
            raise LuaError("ffi: unexpected '%s' in type" % self.peek()[1])
        return ct


# ------------------------------------------------------------------------------------------------------------------------ cdata
def _compat_ptr(d, s, same=False):
    """LuaJIT lj_cconv_compatptr for the pointees d <- s (no cast)"""
    if same:
        if d.const != s.const:
            return False
    else:
        if s.const and not d.const:
            return False                                         # discarded qualifier
        if isinstance(d, Void) or isinstance(s, Void):
            return True
    if type(d) is not type(s):
        return False
    if isinstance(d, Num):
        return d.size == s.size and (d.kind == 'float') == (s.kind == 'float') and (d.kind == 'bool') == (s.kind == 'bool')
    if isinstance(d, Ptr):
        return _compat_ptr(d.to, s.to, True)
    if isinstance(d, Struct):
        return d is s or (d.name == s.name and d.fields is s.fields)
    if isinstance(d, Func):
        return True
    if isinstance(d, Arr):
        return _compat_ptr(d.of, s.of, True)
    return True


class CData(object):
    __slots__ = ()
    lua_type = 'cdata'


class CInt64(CData):
    __slots__ = ('ct', 'val')

    def __init__(self, ct, val):
        self.ct = ct
        bits = 1 << 64
        val = int(val) % bits
        if ct.signed and val >= bits // 2:
            val -= bits
        self.val = val

    def lua_tonumber(self):
        return self.val

    def lua_tostring(self):
        return '%d%s' % (self.val, 'LL' if self.ct.signed else 'ULL')

    def lua_eq(self, a, b):
        x, y = _as_int(a), _as_int(b)
        return x is not None and y is not None and x == y

    def lua_lt(self, a, b):
        return _as_int(a, True) < _as_int(b, True)

    def lua_le(self, a, b):
        return _as_int(a, True) <= _as_int(b, True)

    def lua_arith(self, op, a, b):
        if op == 'unm':
            return CInt64(self.ct, -self.val)
        x, y = _as_int(a, True), _as_int(b, True)
        unsigned = any(isinstance(v, CInt64) and not v.ct.signed for v in (a, b))
        ct = BASE_TYPES['uint64_t'] if unsigned else BASE_TYPES['int64_t']
        if op == '+':
            r = x + y
        elif op == '-':
            r = x - y
        elif op == '*':
            r = x * y
        elif op == '/':
            if y == 0:
                raise LuaError('ffi: 64-bit integer division by zero')
            r = abs(x) // abs(y) * (1 if (x < 0) == (y < 0) else -1)
        elif op == '%':
            r = x - (abs(x) // abs(y) * (1 if (x < 0) == (y < 0) else -1)) * y
        elif op == '^':
            r = x ** y
        else:
            raise LuaError("attempt to perform arithmetic on 'int64_t' and '%s'" % type_name(b))
        return CInt64(ct, r)


def _as_int(v, strict=False):
    if isinstance(v, CInt64):
        return v.val
    if v.__class__ in (int, float):
        return int(v) if float(v).is_integer() or strict else v
    if strict:
        raise LuaError("attempt to perform arithmetic on a %s value and a 64-bit cdata" % type_name(v))
    return None


class CPointer(CData):
    """a pointer VALUE: ct = Ptr(...), val = the address it holds.  `keep` pins Python-owned memory it points into
    (ffi.new buffers reached through ffi.cast / arithmetic are NOT pinned, exactly like LuaJIT -- see CAggregate.decay)."""
    __slots__ = ('ct', 'val', 'keep', '__weakref__')

    def __init__(self, ct, val, keep=None):
        self.ct, self.val, self.keep = ct, int(val or 0), keep

    def lua_tostring(self):
        return 'cdata<%s>: %s' % (self.ct, 'NULL' if self.val == 0 else '0x%012x' % self.val)

    def lua_eq(self, a, b):
        x = a.val if isinstance(a, CPointer) else a.addr if isinstance(a, CAggregate) else 0 if a is None else None
        y = b.val if isinstance(b, CPointer) else b.addr if isinstance(b, CAggregate) else 0 if b is None else None
        return x is not None and y is not None and x == y

    def lua_lt(self, a, b):
        return _addr(a) < _addr(b)

    def lua_le(self, a, b):
        return _addr(a) <= _addr(b)

    def lua_arith(self, op, a, b):
        return _ptr_arith(op, a, b)

    def _elem(self):
        to = self.ct.to
        if to.size is None or isinstance(to, (Func,)):
            raise LuaError("ffi: attempt to index a pointer to the incomplete type '%s'" % to)
        return to

    def lua_index(self, k):
        if k.__class__ is str:
            to = self.ct.to
            if isinstance(to, Struct):
                return _struct_get(to, self.val, k, self.keep)
            raise LuaError("'%s' has no member named '%s'" % (self.ct, k))
        i = _index_int(k)
        to = self._elem()
        if self.val == 0:
            raise LuaError('ffi: NULL pointer dereference (index %d of %s)' % (i, self.ct))
        return load(to, self.val + i * to.size, self.keep)

    def lua_newindex(self, k, v):
        if k.__class__ is str:
            to = self.ct.to
            if isinstance(to, Struct):
                return _struct_set(to, self.val, k, v)
            raise LuaError("'%s' has no member named '%s'" % (self.ct, k))
        i = _index_int(k)
        to = self._elem()
        if to.const:
            raise LuaError("attempt to write to constant location")
        if self.val == 0:
            raise LuaError('ffi: NULL pointer dereference (index %d of %s)' % (i, self.ct))
        store(to, self.val + i * to.size, v)


def _addr(v):
    if isinstance(v, CPointer):
        return v.val
    if isinstance(v, CAggregate):
        return v.addr
    raise LuaError('attempt to compare %s with a pointer' % type_name(v))


def _index_int(k):
    if isinstance(k, CInt64):
        return k.val
    if k.__class__ is int:
        return k
    if k.__class__ is float and k.is_integer():
        return int(k)
    if k.__class__ is float:
        return int(k)                                            # LuaJIT truncates
    raise LuaError("ffi: cdata cannot be indexed with %s" % type_name(k))


def _ptr_arith(op, a, b):
    def as_ptr(v):
        if isinstance(v, CPointer):
            return v
        if isinstance(v, CAggregate) and isinstance(v.ct, Arr):
            return v.decay()
        return None
    pa, pb = as_ptr(a), as_ptr(b)
    if op == '+' and (pa is None) != (pb is None):
        p, n = (pa, b) if pa is not None else (pb, a)
        n = _as_int(n, True) if not isinstance(n, CInt64) else n.val
        if p.ct.to.size is None:
            raise LuaError("ffi: pointer arithmetic on the incomplete type '%s'" % p.ct.to)
        return CPointer(p.ct, p.val + n * p.ct.to.size, p.keep)
    if op == '-' and pa is not None and pb is None:
        n = _as_int(b, True)
        return CPointer(pa.ct, pa.val - n * pa.ct.to.size, pa.keep)
    if op == '-' and pa is not None and pb is not None:
        if pa.ct.to.size != pb.ct.to.size:
            raise LuaError("ffi: attempt to subtract pointers to different types")
        return (pa.val - pb.val) // pa.ct.to.size
    raise LuaError("attempt to perform arithmetic on '%s' and '%s'" % (getattr(a, 'ct', type_name(a)), getattr(b, 'ct', type_name(b))))


_QUARANTINE = []


class CAggregate(CData):
    """an array or struct living in memory: either OWNED (ffi.new: `own` = the ctypes buffer, freed with the cdata) or a
    REFERENCE into someone else's memory (`own` = whatever keeps that alive)."""
    __slots__ = ('ct', 'addr', 'own', '__weakref__')

    def __init__(self, ct, addr, own):
        self.ct, self.addr, self.own = ct, addr, own

    def decay(self):
        return CPointer(Ptr(self.ct.of if isinstance(self.ct, Arr) else self.ct), self.addr, self)

    def lua_tostring(self):
        return 'cdata<%s>: 0x%012x' % (self.ct, self.addr)

    def lua_eq(self, a, b):
        return CPointer.lua_eq(None, a, b)

    def lua_arith(self, op, a, b):
        return _ptr_arith(op, a, b)

    def lua_len(self):
        raise LuaError("attempt to get length of '%s'" % self.ct)

    def lua_index(self, k):
        ct = self.ct
        if isinstance(ct, Struct):
            if k.__class__ is not str:
                raise LuaError("ffi: '%s' cannot be indexed with %s" % (ct, type_name(k)))
            return _struct_get(ct, self.addr, k, self)
        i = _index_int(k)
        if ct.n is not None and not 0 <= i < ct.n:
            raise LuaError("luavm-ffi: index %d is outside '%s' (LuaJIT would read out of bounds)" % (i, ct))
        return load(ct.of, self.addr + i * ct.of.size, self)

    def lua_newindex(self, k, v):
        ct = self.ct
        if isinstance(ct, Struct):
            if k.__class__ is not str:
                raise LuaError("ffi: '%s' cannot be indexed with %s" % (ct, type_name(k)))
            return _struct_set(ct, self.addr, k, v)
        i = _index_int(k)
        if ct.n is not None and not 0 <= i < ct.n:
            raise LuaError("luavm-ffi: index %d is outside '%s' (LuaJIT would write out of bounds)" % (i, ct))
        store(ct.of, self.addr + i * ct.of.size, v)


class CFunc(CData):
    __slots__ = ('ct', 'fn', 'name', 'ffi')

    def __init__(self, ct, fn, name, ffi):
        self.ct, self.fn, self.name, self.ffi = ct, fn, name, ffi

    def lua_tostring(self):
        return 'cdata<%s>: %s' % (self.ct, self.name)

    def lua_call(self, args):
        ct = self.ct
        if len(args) != len(ct.params) and not (ct.vararg and len(args) > len(ct.params)):
            raise LuaError("wrong number of arguments for function call ('%s' takes %d, got %d)" % (self.name, len(ct.params), len(args)))
        raw, keep = [], []
        for i, p in enumerate(ct.params):
            try:
                raw.append(to_c(p, args[i], keep))
            except LuaError as e:
                raise LuaError("bad argument #%d to '%s' (%s)" % (i + 1, self.name, e.value))
        for a in args[len(ct.params):]:
            raw.append(a)
        if self.ffi.trace is not None:
            self.ffi.trace.append((self.name, tuple(raw)))
        r = self.fn(*raw)
        del keep
        ret = ct.ret
        if isinstance(ret, Void):
            return None
        if isinstance(ret, Ptr):
            return CPointer(ret, r or 0)
        if isinstance(ret, Num):
            return from_c_num(ret, r)
        raise LuaError('ffi: unsupported return type %s' % ret)


def from_c_num(ct, v):
    if ct.kind == 'bool':
        return bool(v)
    if ct.kind == 'float':
        return float(v)
    if ct.size == 8:
        return CInt64(ct, v)
    return int(v)


def load(ct, addr, keep=None):
    """C -> Lua read of an object of type ct at addr"""
    if isinstance(ct, Num):
        return from_c_num(ct, ct.ctype.from_address(addr).value)
    if isinstance(ct, Ptr):
        return CPointer(ct, ctypes.c_void_p.from_address(addr).value or 0)
    if isinstance(ct, (Struct, Arr)):
        return CAggregate(ct, addr, keep)                        # a reference, not a copy
    raise LuaError('ffi: cannot read a value of type %s' % ct)


def to_c(ct, v, keep, cast=False):
    """Lua -> C conversion of v for a destination of type ct; returns the raw value (int / float / address)"""
    if isinstance(ct, Num):
        if v.__class__ is bool:
            if ct.kind == 'bool' or cast:
                return int(v)
            raise LuaError("cannot convert 'boolean' to '%s'" % ct)
        if v.__class__ in (int, float):
            if ct.kind == 'float':
                return float(v)
            if ct.kind == 'bool':
                return v != 0
            if v != v or v in (float('inf'), float('-inf')):
                return 0
            iv = int(v)                                          # truncation, like (int)double
            bits = 1 << (8 * ct.size)
            iv %= bits
            if ct.signed and iv >= bits // 2:
                iv -= bits
            return iv
        if isinstance(v, CInt64):
            return float(v.val) if ct.kind == 'float' else CInt64(ct, v.val).val if ct.size == 8 else to_c(ct, v.val, keep)
        if isinstance(v, CPointer) and cast and ct.kind == 'int':
            return v.val
        raise LuaError("cannot convert '%s' to '%s'" % (_tname(v), ct))
    if isinstance(ct, Ptr):
        if v is None:
            return None
        if isinstance(v, CPointer):
            if not cast and not _compat_ptr(ct.to, v.ct.to):
                raise LuaError("cannot convert '%s' to '%s'" % (v.ct, ct))
            keep.append(v)
            return v.val or None
        if isinstance(v, CAggregate):
            src = v.ct.of if isinstance(v.ct, Arr) else v.ct
            if not cast and not _compat_ptr(ct.to, src):
                raise LuaError("cannot convert '%s' to '%s'" % (v.ct, ct))
            keep.append(v)
            return v.addr
        if v.__class__ is str:
            cchar = BASE_TYPES['char'].with_const()
            if not cast and not _compat_ptr(ct.to, cchar):
                raise LuaError("cannot convert 'string' to '%s'" % ct)
            buf = ctypes.create_string_buffer(v.encode('latin-1'), len(v) + 1)
            keep.append(buf)
            return ctypes.addressof(buf)
        if isinstance(v, CFunc) and (cast or isinstance(ct.to, (Func, Void))):
            return ctypes.cast(v.fn, ctypes.c_void_p).value
        if cast and (v.__class__ in (int, float) or isinstance(v, CInt64)):
            return int(v.val if isinstance(v, CInt64) else v) or None
        raise LuaError("cannot convert '%s' to '%s'" % (_tname(v), ct))
    raise LuaError("cannot convert '%s' to '%s'" % (_tname(v), ct))


def _tname(v):
    if isinstance(v, CData):
        return str(getattr(v, 'ct', 'cdata'))
    return type_name(v)


def store(ct, addr, v):
    if ct.const:
        raise LuaError('attempt to write to constant location')
    if isinstance(ct, Num):
        ct.ctype.from_address(addr).value = to_c(ct, v, [])
        return
    if isinstance(ct, Ptr):
        keep = []
        ctypes.c_void_p.from_address(addr).value = to_c(ct, v, keep)
        # NOT pinned: a pointer stored into C memory does not keep its target alive (same as LuaJIT)
        return
    if isinstance(ct, Struct) and isinstance(v, CAggregate) and v.ct is ct:
        ctypes.memmove(addr, v.addr, ct.size)
        return
    if isinstance(ct, Arr) and v.__class__ is str and ct.of.size == 1:
        data = v.encode('latin-1')[:ct.size]
        ctypes.memmove(addr, data, len(data))
        if len(data) < ct.size:
            ctypes.memset(addr + len(data), 0, 1)
        return
    raise LuaError("cannot convert '%s' to '%s'" % (_tname(v), ct))


def _struct_get(st, addr, name, keep):
    if st.fields is None:
        raise LuaError("ffi: attempt to index the incomplete type '%s'" % st)
    f = st.index.get(name)
    if f is None:
        raise LuaError("'%s' has no member named '%s'" % (st, name))
    if addr == 0:
        raise LuaError("ffi: NULL pointer dereference (field '%s')" % name)
    return load(f[0], addr + f[1], keep)


def _struct_set(st, addr, name, v):
    if st.fields is None:
        raise LuaError("ffi: attempt to index the incomplete type '%s'" % st)
    f = st.index.get(name)
    if f is None:
        raise LuaError("'%s' has no member named '%s'" % (st, name))
    if st.const:
        raise LuaError('attempt to write to constant location')
    try:
        store(f[0], addr + f[1], v)
    except LuaError as e:
        raise LuaError("%s (field '%s')" % (e.value, name))


class CTypeObj(CData):
    """result of ffi.typeof: callable constructor"""
    __slots__ = ('ffi', 'ct')

    def __init__(self, ffi, ct):
        self.ffi, self.ct = ffi, ct

    def lua_tostring(self):
        return 'ctype<%s>' % self.ct

    def lua_call(self, args):
        return self.ffi.new(self.ct, *args)


class CLib(object):
    """ffi.load result / ffi.C: symbols resolved on first use against the cdef'ed declarations"""
    lua_type = 'userdata'

    def __init__(self, ffi, handle, name):
        self.ffi, self.handle, self.name, self.cache = ffi, handle, name, {}

    def lua_tostring(self):
        return 'library: %s' % self.name

    def lua_index(self, k):
        if k in self.cache:
            return self.cache[k]
        if k in self.ffi.constants:
            return self.ffi.constants[k]
        ct = self.ffi.decls.get(k)
        if ct is None:
            raise LuaError("missing declaration for symbol '%s'" % k)
        if not isinstance(ct, Func):
            raise LuaError('luavm-ffi: only function symbols are supported (%s)' % k)
        try:
            fn = getattr(self.handle, k)
        except AttributeError:
            raise LuaError("cannot resolve symbol '%s': undefined symbol in %s" % (k, self.name))
        if isinstance(fn, ctypes._CFuncPtr):
            fn.argtypes = [_ctypes_of(p) for p in ct.params] if not ct.vararg else None
            fn.restype = _ctypes_of(ct.ret)
        cf = CFunc(ct, fn, k, self.ffi)
        self.cache[k] = cf
        return cf

    def lua_newindex(self, k, v):
        raise LuaError('attempt to write to a library namespace')


def _ctypes_of(ct):
    if isinstance(ct, Void):
        return None
    if isinstance(ct, Num):
        return ct.ctype
    if isinstance(ct, Ptr):
        return ctypes.c_void_p
    raise LuaError('luavm-ffi: unsupported type in a prototype: %s' % ct)


class FFI(object):
    """one per VM: the declaration tables + the Lua-visible `ffi` module (self.module)"""

    def __init__(self, vm, loader=None):
        self.vm = vm
        self.typedefs, self.structs, self.decls, self.constants = {}, {}, {}, {}
        self.loader = loader or (lambda path, glob: ctypes.CDLL(path, mode=ctypes.RTLD_GLOBAL if glob else ctypes.RTLD_LOCAL))
        self.trace = None                # list -> every C call appends (name, raw args)
        self.finalizers = []
        self._types = {}
        m = self.module = LuaTable()
        for name in ('cdef', 'load', 'new', 'cast', 'typeof', 'sizeof', 'string', 'gc', 'copy', 'fill', 'istype', 'errno', 'abi'):
            m.set(name, getattr(self, 'l_' + name))
        m.set('os', 'Linux')
        m.set('arch', 'x64')
        m.set('C', CLib(self, ctypes.CDLL(None), 'C'))
        vm.make_int64 = lambda v, unsigned: CInt64(BASE_TYPES['uint64_t' if unsigned else 'int64_t'], v)
        vm.at_close.append(self.run_finalizers)

    # ---- types
    def ctype(self, spec):
        if isinstance(spec, CT):
            return spec
        if isinstance(spec, CTypeObj):
            return spec.ct
        if isinstance(spec, (CPointer, CAggregate, CInt64)):
            return spec.ct
        if spec.__class__ is not str:
            raise LuaError("bad argument #1 (C type expected, got %s)" % type_name(spec))
        ct = self._types.get(spec)
        if ct is None:
            ct = CParser(self, spec).abstract_type()
            self._types[spec] = ct
        return ct

    def pointer_to(self, type_name_, addr, keep=None):
        """helper for host objects (tensor:data()): a `type_name_ *` cdata"""
        return CPointer(Ptr(self.ctype(type_name_)), addr, keep)

    # ---- Lua-visible functions
    def l_cdef(self, src=None, *_):
        if src.__class__ is not str:
            raise LuaError("bad argument #1 to 'cdef' (string expected, got %s)" % type_name(src))
        CParser(self, src).declarations()
        self._types.clear()

    def l_load(self, name=None, glob=None, *_):
        if name.__class__ is not str:
            raise LuaError("bad argument #1 to 'load' (string expected, got %s)" % type_name(name))
        try:
            return CLib(self, self.loader(name, bool(glob)), name)
        except OSError as e:
            raise LuaError('%s' % e)

    def new(self, ct, *args):
        ct = self.ctype(ct)
        args = list(args)
        if isinstance(ct, Arr) and ct.n is None:
            if not args or _as_int(args[0]) is None:
                raise LuaError("bad argument #2 to 'new' (number expected for the VLA size)")
            n = int(_as_int(args.pop(0)))
            if n < 0:
                raise LuaError('ffi.new: negative array size')
            ct = Arr(ct.of, n)
        if ct.size is None:
            raise LuaError("size of C type is unknown or too large ('%s')" % ct)
        if isinstance(ct, (Arr, Struct)):
            buf = ctypes.create_string_buffer(max(ct.size, 1))
            cd = CAggregate(ct, ctypes.addressof(buf), buf)
            if args:
                self._init(ct, cd.addr, args)
            return cd
        if isinstance(ct, Ptr):
            return CPointer(ct, to_c(ct, args[0], []) if args else 0)
        if isinstance(ct, Num):
            v = to_c(ct, args[0], []) if args else 0
            return CInt64(ct, v) if (ct.size == 8 and ct.kind == 'int') else from_c_num(ct, v)
        raise LuaError("ffi.new: cannot create a '%s'" % ct)

    def _init(self, ct, addr, args):
        if isinstance(ct, Arr):
            if len(args) == 1 and args[0].__class__ is LuaTable:
                vals = [args[0].get(i + 1) for i in range(args[0].length())]
                if len(vals) > ct.n:
                    raise LuaError('too many initializers for %s' % ct)
                if len(vals) == 1:
                    vals = vals * ct.n
            elif len(args) == 1 and args[0].__class__ is str and ct.of.size == 1:
                return store(ct, addr, args[0])
            elif len(args) == 1:
                vals = [args[0]] * ct.n                          # a single initialiser is replicated
            else:
                if len(args) > ct.n:
                    raise LuaError('too many initializers for %s' % ct)
                vals = args
            for i, v in enumerate(vals):
                if isinstance(ct.of, (Arr, Struct)) and v.__class__ is LuaTable:
                    self._init(ct.of, addr + i * ct.of.size, [v])
                else:
                    store(ct.of, addr + i * ct.of.size, v)
            return
        if isinstance(ct, Struct):
            if len(args) == 1 and args[0].__class__ is LuaTable:
                t = args[0]
                if t.length() > 0:
                    args = [t.get(i + 1) for i in range(t.length())]
                else:
                    for k, v in t.items():
                        _struct_set(ct, addr, k, v)
                    return
            elif len(args) == 1 and isinstance(args[0], CAggregate):
                return store(ct, addr, args[0])
            if len(args) > len(ct.fields):
                raise LuaError('too many initializers for %s' % ct)
            for (fname, fct, off), v in zip(ct.fields, args):
                store(fct, addr + off, v)

    def l_new(self, ct=None, *args):
        return self.new(ct, *args)

    def l_cast(self, ct=None, v=None, *_):
        ct = self.ctype(ct)
        if isinstance(ct, Ptr):
            keep = []
            raw = to_c(ct, v, keep, cast=True)
            pin = None
            if v.__class__ is str:
                pin = keep[0]                                    # LuaJIT: valid while the string is alive; luavm pins a copy
            return CPointer(ct, raw or 0, pin)
        if isinstance(ct, Num):
            raw = to_c(ct, v, [], cast=True)
            return CInt64(ct, raw) if (ct.size == 8 and ct.kind == 'int') else from_c_num(ct, raw)
        raise LuaError("ffi.cast: unsupported target type '%s'" % ct)

    def l_typeof(self, ct=None, *_):
        return CTypeObj(self, self.ctype(ct))

    def l_sizeof(self, ct=None, n=None, *_):
        ct = self.ctype(ct)
        if isinstance(ct, Arr) and ct.n is None:
            return ct.of.size * int(n)
        return ct.size

    def l_string(self, p=None, n=None, *_):
        if p.__class__ is str:
            return p if n is None else p[:int(n)]
        if isinstance(p, CAggregate):
            addr, limit = p.addr, p.ct.size
        elif isinstance(p, CPointer):
            addr, limit = p.val, None
        else:
            raise LuaError("bad argument #1 to 'string' (cannot convert '%s' to 'const char *')" % _tname(p))
        if addr == 0:
            raise LuaError("bad argument #1 to 'string' (NULL pointer)")
        if n is not None:
            n = int(_as_int(n))
            if limit is not None and n > limit:
                raise LuaError('luavm-ffi: ffi.string length %d exceeds the %d-byte buffer' % (n, limit))
            return ctypes.string_at(addr, n).decode('latin-1')
        s = ctypes.string_at(addr)
        if limit is not None and len(s) > limit:
            s = s[:limit]
        return s.decode('latin-1')

    def l_gc(self, cd=None, fin=None, *_):
        if not isinstance(cd, (CPointer, CAggregate)):
            raise LuaError("bad argument #1 to 'gc' (cdata expected, got %s)" % type_name(cd))
        for rec in list(self.finalizers):
            if rec[0]() is cd:
                rec[1].detach()
                self.finalizers.remove(rec)
        if fin is not None:
            snapshot = CPointer(cd.ct, cd.val) if isinstance(cd, CPointer) else None

            def run(vm_ref=weakref.ref(self.vm)):
                if snapshot is not None:
                    call(fin, [snapshot])
            f = weakref.finalize(cd, run)
            f.atexit = False
            self.finalizers.append((weakref.ref(cd), f))
        return cd

    def run_finalizers(self):
        for _ref, f in reversed(self.finalizers):
            f()
        self.finalizers = []

    def l_copy(self, dst=None, src=None, n=None, *_):
        d = _addr(dst)
        if src.__class__ is str:
            data = src.encode('latin-1')
            n = len(data) + 1 if n is None else int(n)
            ctypes.memmove(d, data + b'\0', min(n, len(data) + 1))
            return
        ctypes.memmove(d, _addr(src), int(_as_int(n)))

    def l_fill(self, dst=None, n=None, c=0, *_):
        ctypes.memset(_addr(dst), int(c or 0), int(_as_int(n)))

    def l_istype(self, ct=None, v=None, *_):
        ct = self.ctype(ct)
        return isinstance(v, CData) and hasattr(v, 'ct') and str(v.ct).replace('const ', '') == str(ct).replace('const ', '')

    def l_errno(self, *_):
        return ctypes.get_errno()

    def l_abi(self, what=None, *_):
        return what in ('64bit', 'le', 'fpu', 'hardfp')


def install(vm, loader=None):
    ffi = FFI(vm, loader)
    vm.preload.set('ffi', lambda *_: ffi.module)
    vm.ffi = ffi
    return ffi
