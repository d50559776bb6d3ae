"""The Lua 5.1 evaluator: AST (parse.py) -> Python closures, plus the base / string / table / math / os / io libraries.

Values: nil = None, booleans = bool, numbers = int | float (one Lua type; ints stay exact like doubles below 2^53), strings = str
with one char per byte, tables = LuaTable, functions = LuaFunction | any Python callable, userdata / cdata = any Python object with
the `lua_*` protocol (lua_index, lua_newindex, lua_call, lua_arith, lua_eq, lua_len, lua_tostring, lua_type).
Every local lives in a one-element list (a cell) created when its `local` statement runs, so closures capture variables per
iteration exactly like Lua's upvalues.  A block returns None (fell through), BREAK, or a list (the values of a `return`).
"""
import math
import os
import re
import sys
import time
from functools import cmp_to_key

from .parse import parse

BREAK = object()
_TRUE_KEY, _FALSE_KEY = object(), object()       # bool keys must not collide with 1 / 0 in a Python dict


class LuaError(Exception):
    def __init__(self, value, traceback=None, level=None):
        Exception.__init__(self, value)
        self.value = value
        self.tb = traceback or []
        self.level = level           # error(msg, level): call sites still to unwind before the position is prefixed

    def __str__(self):
        v = self.value
        s = v if isinstance(v, str) else tostring(v)
        if self.tb:
            s += '\nstack traceback:\n' + '\n'.join('\t%s:%d: in %s' % t for t in self.tb[:30])
        return s


class LuaTable(object):
    __slots__ = ('arr', 'hash', 'meta', '__weakref__')

    def __init__(self):
        self.arr = []          # keys 1 .. len(arr)
        self.hash = {}
        self.meta = None

    def get(self, k):
        c = k.__class__
        if c is int:
            if 0 < k <= len(self.arr):
                return self.arr[k - 1]
            return self.hash.get(k)
        if c is str:
            return self.hash.get(k)
        if c is float:
            if k.is_integer():
                return self.get(int(k))
            return self.hash.get(k)
        if c is bool:
            return self.hash.get(_TRUE_KEY if k else _FALSE_KEY)
        if k is None:
            return None
        return self.hash.get(k)

    def set(self, k, v):
        c = k.__class__
        if c is float:
            if k.is_integer():
                k, c = int(k), int
            elif k != k:
                raise LuaError('table index is NaN')
        if c is int:
            n = len(self.arr)
            if 0 < k <= n:
                self.arr[k - 1] = v
                if v is None and k == n:
                    while self.arr and self.arr[-1] is None:
                        self.arr.pop()
                return
            if k == n + 1:
                if v is None:
                    self.hash.pop(k, None)
                    return
                self.arr.append(v)
                if self.hash:
                    self.hash.pop(k, None)
                    nxt = k + 1
                    while nxt in self.hash:
                        self.arr.append(self.hash.pop(nxt))
                        nxt += 1
                return
        elif c is bool:
            k = _TRUE_KEY if k else _FALSE_KEY
        elif k is None:
            raise LuaError('table index is nil')
        if v is None:
            self.hash.pop(k, None)
        else:
            self.hash[k] = v

    def length(self):
        return len(self.arr)

    def keys(self):
        ks = [i + 1 for i, v in enumerate(self.arr) if v is not None]
        for k in self.hash:
            ks.append(True if k is _TRUE_KEY else False if k is _FALSE_KEY else k)
        return ks

    def next(self, k):
        ks = self.keys()
        if k is None:
            i = 0
        else:
            try:
                i = ks.index(k) + 1
            except ValueError:
                raise LuaError("invalid key to 'next'")
        if i >= len(ks):
            return None, None
        return ks[i], self.get(ks[i])

    # convenience for Python-side code
    def __getitem__(self, k):
        return self.get(k)

    def __setitem__(self, k, v):
        self.set(k, v)

    def items(self):
        return [(k, self.get(k)) for k in self.keys()]


class Proto(object):
    __slots__ = ('name', 'nparams', 'is_vararg', 'nslots', 'body', 'upvals', 'chunk', 'line')


class Frame(object):
    __slots__ = ('s', 'u', 'va')


class LuaFunction(object):
    __slots__ = ('proto', 'up', 'vm', '__weakref__')

    def __init__(self, proto, up, vm):
        self.proto, self.up, self.vm = proto, up, vm

    def __call__(self, *args):                   # Python-side convenience: returns the list of results
        return call(self, list(args))

    def __repr__(self):
        return 'function: %s (%s:%d)' % (self.proto.name, self.proto.chunk, self.proto.line)


def type_name(v):
    if v is None:
        return 'nil'
    c = v.__class__
    if c is bool:
        return 'boolean'
    if c is int or c is float:
        return 'number'
    if c is str:
        return 'string'
    if c is LuaTable:
        return 'table'
    if c is LuaFunction:
        return 'function'
    t = getattr(v, 'lua_type', None)
    if t is not None:
        return t
    if callable(v):
        return 'function'
    return 'userdata'


def fmt_number(v):
    if v.__class__ is int:
        return str(v)
    if v != v:
        return 'nan'
    if v in (float('inf'), float('-inf')):
        return 'inf' if v > 0 else '-inf'
    if v.is_integer() and abs(v) < 1e15:
        return str(int(v))
    return '%.14g' % v


def tostring(v):
    if v is None:
        return 'nil'
    c = v.__class__
    if c is str:
        return v
    if c is bool:
        return 'true' if v else 'false'
    if c is int or c is float:
        return fmt_number(v)
    if c is LuaTable:
        if v.meta is not None:
            h = v.meta.get('__tostring')
            if h is not None:
                return first(call(h, [v]))
        return 'table: 0x%08x' % (id(v) & 0xffffffff)
    f = getattr(v, 'lua_tostring', None)
    if f is not None:
        return f()
    if c is LuaFunction:
        return 'function: 0x%08x' % (id(v) & 0xffffffff)
    if callable(v):
        return 'function: builtin: %s' % getattr(v, '__name__', '?')
    return 'userdata: 0x%08x' % (id(v) & 0xffffffff)


_NUMRE = re.compile(r'^\s*[-+]?(0[xX][0-9a-fA-F]+|(\d+\.?\d*|\.\d+)([eE][-+]?\d+)?)\s*$')


def tonumber(v, base=None):
    c = v.__class__
    if base is None or base == 10:
        if c is int or c is float:
            return v
        if c is str:
            if not _NUMRE.match(v):
                return None
            s = v.strip()
            try:
                if s.lower().lstrip('+-').startswith('0x'):
                    return int(s, 16)
                f = float(s)
                return int(f) if f.is_integer() and abs(f) < 2 ** 53 and 'e' not in s.lower() and '.' not in s else f
            except ValueError:
                return None
        f = getattr(v, 'lua_tonumber', None)
        return f() if f is not None else None
    try:
        return int(tostring(v).strip(), int(base))
    except ValueError:
        return None


def first(vals):
    return vals[0] if vals else None


def truthy(v):
    return v is not None and v is not False


# ---------------------------------------------------------------------------------------------------------- calls / metamethods
def call(f, args):
    """-> list of results"""
    c = f.__class__
    if c is LuaFunction:
        proto = f.proto
        fr = Frame()
        s = fr.s = [None] * proto.nslots
        fr.u = f.up
        np_, n = proto.nparams, len(args)
        if n >= np_:
            for i in range(np_):
                s[i] = [args[i]]
            fr.va = args[np_:] if proto.is_vararg else None
        else:
            for i in range(n):
                s[i] = [args[i]]
            for i in range(n, np_):
                s[i] = [None]
            fr.va = [] if proto.is_vararg else None
        r = proto.body(fr)
        return r if r.__class__ is list else []
    if c is LuaTable:
        h = f.meta.get('__call') if f.meta is not None else None
        if h is None:
            raise LuaError('attempt to call a table value')
        return call(h, [f] + args)
    lc = getattr(f, 'lua_call', None)
    if lc is not None:
        r = lc(args)
    elif callable(f):
        r = f(*args)
    else:
        raise LuaError('attempt to call a %s value' % type_name(f))
    if r is None:
        return [None]                  # a Python builtin that returns None returns ONE nil (tonumber, string.match, os.getenv, next ...)
    if r.__class__ is tuple:
        return list(r)
    if r.__class__ is list:
        return r
    return [r]


def index(o, k):
    c = o.__class__
    if c is LuaTable:
        v = o.get(k)
        if v is None and o.meta is not None:
            h = o.meta.hash.get('__index')
            if h is not None:
                if h.__class__ is LuaTable:
                    return index(h, k)
                return first(call(h, [o, k]))
        return v
    if c is str:
        return STRING_LIB.get(k)
    li = getattr(o, 'lua_index', None)
    if li is not None:
        return li(k)
    raise LuaError('attempt to index a %s value' % type_name(o))


def setindex(o, k, v):
    c = o.__class__
    if c is LuaTable:
        if o.meta is not None and o.get(k) is None:
            h = o.meta.hash.get('__newindex')
            if h is not None:
                if h.__class__ is LuaTable:
                    return setindex(h, k, v)
                call(h, [o, k, v])
                return
        o.set(k, v)
        return
    ls = getattr(o, 'lua_newindex', None)
    if ls is not None:
        ls(k, v)
        return
    raise LuaError('attempt to index a %s value' % type_name(o))


def _meta_of(v):
    if v.__class__ is LuaTable:
        return v.meta
    return None


_ARITH_EVENT = {'+': '__add', '-': '__sub', '*': '__mul', '/': '__div', '%': '__mod', '^': '__pow', '..': '__concat'}


def arith_slow(op, a, b):
    for v in (a, b):
        m = _meta_of(v)
        if m is not None:
            h = m.get(_ARITH_EVENT[op])
            if h is not None:
                return first(call(h, [a, b]))
        la = getattr(v, 'lua_arith', None)
        if la is not None:
            return la(op, a, b)
    if op != '..':
        x, y = (tonumber(a) if a.__class__ is str else None), (tonumber(b) if b.__class__ is str else None)
        x = a if a.__class__ in (int, float) else x
        y = b if b.__class__ in (int, float) else y
        if x is not None and y is not None:
            return arith(op, x, y)
        bad = b if x is not None else a
        raise LuaError('attempt to perform arithmetic on a %s value' % type_name(bad))
    bad = b if a.__class__ in (str, int, float) else a
    raise LuaError('attempt to concatenate a %s value' % type_name(bad))


def arith(op, a, b):
    ca, cb = a.__class__, b.__class__
    if (ca is int or ca is float) and (cb is int or cb is float):
        if op == '+':
            return a + b
        if op == '-':
            return a - b
        if op == '*':
            return a * b
        if op == '/':
            if b == 0:
                if a == 0 or a != a:
                    return float('nan')
                return float('inf') if (a > 0) == (math.copysign(1.0, b) > 0) else float('-inf')
            r = a / b
            return r
        if op == '%':
            if b == 0:
                return float('nan')
            return a - math.floor(a / b) * b
        if op == '^':
            try:
                if ca is int and cb is int and b >= 0 and abs(a) ** b < 2 ** 53:
                    return a ** b
                return float(a) ** float(b)
            except (OverflowError, ZeroDivisionError):
                return float('inf')
    return arith_slow(op, a, b)


def concat(a, b):
    ca, cb = a.__class__, b.__class__
    if (ca is str or ca is int or ca is float) and (cb is str or cb is int or cb is float):
        return (a if ca is str else fmt_number(a)) + (b if cb is str else fmt_number(b))
    return arith_slow('..', a, b)


def lua_eq(a, b):
    if a is b:
        return True
    ca, cb = a.__class__, b.__class__
    if ca is bool or cb is bool:
        return False                              # `a is b` covered equal booleans
    if (ca is int or ca is float) and (cb is int or cb is float):
        return a == b
    if ca is str and cb is str:
        return a == b
    if ca is LuaTable and cb is LuaTable:
        if a.meta is not None and b.meta is not None:
            h = a.meta.get('__eq')
            if h is not None and h is b.meta.get('__eq'):
                return truthy(first(call(h, [a, b])))
        return False
    for v in (a, b):
        f = getattr(v, 'lua_eq', None)
        if f is not None:
            return f(a, b)
    return False


def lua_lt(a, b):
    ca, cb = a.__class__, b.__class__
    if (ca is int or ca is float) and (cb is int or cb is float):
        return a < b
    if ca is str and cb is str:
        return a < b
    for v in (a, b):
        m = _meta_of(v)
        if m is not None and m.get('__lt') is not None:
            return truthy(first(call(m.get('__lt'), [a, b])))
        f = getattr(v, 'lua_lt', None)
        if f is not None:
            return f(a, b)
    raise LuaError('attempt to compare %s with %s' % (type_name(a), type_name(b)))


def lua_le(a, b):
    ca, cb = a.__class__, b.__class__
    if (ca is int or ca is float) and (cb is int or cb is float):
        return a <= b
    if ca is str and cb is str:
        return a <= b
    for v in (a, b):
        m = _meta_of(v)
        if m is not None and m.get('__le') is not None:
            return truthy(first(call(m.get('__le'), [a, b])))
        f = getattr(v, 'lua_le', None)
        if f is not None:
            return f(a, b)
    for v in (a, b):
        m = _meta_of(v)
        if m is not None and m.get('__lt') is not None:
            return not truthy(first(call(m.get('__lt'), [b, a])))
    raise LuaError('attempt to compare %s with %s' % (type_name(a), type_name(b)))


def lua_len(v):
    c = v.__class__
    if c is str:
        return len(v)
    if c is LuaTable:
        return len(v.arr)
    f = getattr(v, 'lua_len', None)
    if f is not None:
        return f()
    raise LuaError('attempt to get length of a %s value' % type_name(v))


def lua_unm(v):
    c = v.__class__
    if c is int or c is float:
        return -v
    if c is str and tonumber(v) is not None:
        return -tonumber(v)
    m = _meta_of(v)
    if m is not None and m.get('__unm') is not None:
        return first(call(m.get('__unm'), [v, v]))
    f = getattr(v, 'lua_arith', None)
    if f is not None:
        return f('unm', v, None)
    raise LuaError('attempt to perform arithmetic on a %s value' % type_name(v))


# ---------------------------------------------------------------------------------------------------------------- the compiler
class FuncState(object):
    def __init__(self, parent):
        self.parent = parent
        self.blocks = [{}]
        self.nslots = 0
        self.upvals = []          # (from_parent_local: bool, index)
        self.upindex = {}
        self.is_vararg = False

    def declare(self, name):
        idx = self.nslots
        self.nslots += 1
        self.blocks[-1][name] = idx
        return idx

    def find_local(self, name):
        for b in reversed(self.blocks):
            if name in b:
                return b[name]
        return None

    def resolve(self, name):
        idx = self.find_local(name)
        if idx is not None:
            return ('local', idx)
        if name in self.upindex:
            return ('up', self.upindex[name])
        if self.parent is None:
            return ('global', name)
        r = self.parent.resolve(name)
        if r[0] == 'global':
            return r
        self.upvals.append((r[0] == 'local', r[1]))
        self.upindex[name] = len(self.upvals) - 1
        return ('up', len(self.upvals) - 1)


def _describe(node):
    if node[0] == 'name':
        return " (variable '%s')" % node[1]
    if node[0] == 'index' and node[2][0] == 'str':
        return " (field '%s')" % node[2][1]
    if node[0] == 'method':
        return " (method '%s')" % node[2]
    return ''


class Compiler(object):
    def __init__(self, vm, chunkname):
        self.vm = vm
        self.chunk = chunkname
        self.G = vm.globals

    def err(self, e, line, what='?'):
        e.tb.append((self.chunk, line, what))
        return e

    # ---- expressions
    def expr(self, node, fs):
        kind = node[0]
        if kind == 'nil':
            return lambda fr: None
        if kind == 'true':
            return lambda fr: True
        if kind == 'false':
            return lambda fr: False
        if kind == 'num' or kind == 'str':
            v = node[1]
            return lambda fr: v
        if kind == 'num64':
            v = self.vm.make_int64(*node[1])
            return lambda fr: v
        if kind == 'paren':
            return self.expr(node[1], fs)
        if kind == 'name':
            r = fs.resolve(node[1])
            if r[0] == 'local':
                i = r[1]
                return lambda fr: fr.s[i][0]
            if r[0] == 'up':
                i = r[1]
                return lambda fr: fr.u[i][0]
            gh, name = self.G.hash, node[1]
            return lambda fr: gh.get(name)
        if kind == 'vararg':
            if not fs.is_vararg:
                raise LuaError("%s:%d: cannot use '...' outside a vararg function" % (self.chunk, node[1]))
            return lambda fr: fr.va[0] if fr.va else None
        if kind == 'index':
            obj, key, line, chunk = self.expr(node[1], fs), self.expr(node[2], fs), node[3], self.chunk
            desc = _describe(node[1])

            def ev_index(fr):
                o = obj(fr)
                if o.__class__ is LuaTable:
                    k = key(fr)
                    v = o.get(k)
                    if v is None and o.meta is not None:
                        return index(o, k)
                    return v
                try:
                    return index(o, key(fr))
                except LuaError as e:
                    if not e.tb:
                        e.value = '%s:%d: %s%s' % (chunk, line, e.value, desc) if isinstance(e.value, str) else e.value
                        e.tb.append((chunk, line, 'index'))
                    raise
            return ev_index
        if kind == 'call' or kind == 'method':
            m = self.multi(node, fs)
            return lambda fr: (m(fr) or (None,))[0]
        if kind == 'func':
            return self.function(node, fs)
        if kind == 'and':
            a, b = self.expr(node[1], fs), self.expr(node[2], fs)

            def ev_and(fr):
                v = a(fr)
                if v is None or v is False:
                    return v
                return b(fr)
            return ev_and
        if kind == 'or':
            a, b = self.expr(node[1], fs), self.expr(node[2], fs)

            def ev_or(fr):
                v = a(fr)
                if v is None or v is False:
                    return b(fr)
                return v
            return ev_or
        if kind == 'bin':
            return self.binop(node, fs)
        if kind == 'un':
            op, a, line, chunk = node[1], self.expr(node[2], fs), node[3], self.chunk
            desc = _describe(node[2])
            fn = {'-': lua_unm, '#': lua_len, 'not': None}[op]
            if op == 'not':
                return lambda fr: not truthy(a(fr))

            def ev_un(fr):
                try:
                    return fn(a(fr))
                except LuaError as e:
                    if not e.tb:
                        e.value = '%s:%d: %s%s' % (chunk, line, e.value, desc) if isinstance(e.value, str) else e.value
                        e.tb.append((chunk, line, op))
                    raise
            return ev_un
        if kind == 'table':
            return self.table(node, fs)
        raise LuaError('compiler: unknown expression %r' % (kind,))

    def binop(self, node, fs):
        op, a, b, line, chunk = node[1], self.expr(node[2], fs), self.expr(node[3], fs), node[4], self.chunk
        da, db = _describe(node[2]), _describe(node[3])

        def guard(fn):
            def guarded(x, y):
                try:
                    return fn(x, y)
                except LuaError as e:
                    if not e.tb:
                        if isinstance(e.value, str):
                            bad = da if (x is None or x.__class__ in (bool, LuaTable)) else db
                            e.value = '%s:%d: %s%s' % (chunk, line, e.value, bad)
                        e.tb.append((chunk, line, op))
                    raise
            return guarded

        def wrap(fn):
            g = guard(fn)
            return lambda fr: g(a(fr), b(fr))
        if op == '+':
            slow = guard(lambda x, y: arith('+', x, y))

            def ev_add(fr):
                x, y = a(fr), b(fr)
                cx, cy = x.__class__, y.__class__
                if (cx is int or cx is float) and (cy is int or cy is float):
                    return x + y
                return slow(x, y)
            return ev_add
        if op in ('-', '*', '/', '%', '^'):
            return wrap(lambda x, y: arith(op, x, y))
        if op == '..':
            return wrap(concat)
        if op == '==':
            return lambda fr: lua_eq(a(fr), b(fr))
        if op == '~=':
            return lambda fr: not lua_eq(a(fr), b(fr))
        if op == '<':
            return wrap(lua_lt)
        if op == '<=':
            return wrap(lua_le)
        if op == '>':
            return wrap(lambda x, y: lua_lt(y, x))
        if op == '>=':
            return wrap(lambda x, y: lua_le(y, x))
        raise LuaError('compiler: unknown operator %r' % op)

    def table(self, node, fs):
        items = node[1]
        pos, hashed = [], []
        for i, (kind, k, v) in enumerate(items):
            if kind == 'pos':
                last = (i == len(items) - 1) and v[0] in ('call', 'method', 'vararg')
                pos.append((self.multi(v, fs) if last else self.expr(v, fs), last))
            else:
                hashed.append((self.expr(k, fs), self.expr(v, fs)))

        def ev_table(fr):
            t = LuaTable()
            for k, v in hashed:
                kk = k(fr)
                if kk is None:
                    raise LuaError('table index is nil')
                t.set(kk, v(fr))
            arr = t.arr
            if arr or t.hash:
                n = 0
                for f, is_multi in pos:
                    if is_multi:
                        for x in f(fr):
                            n += 1
                            t.set(n, x)
                    else:
                        n += 1
                        t.set(n, f(fr))
            else:
                for f, is_multi in pos:
                    if is_multi:
                        arr.extend(f(fr))
                    else:
                        arr.append(f(fr))
                while arr and arr[-1] is None:
                    arr.pop()
            return t
        return ev_table

    def multi(self, node, fs):
        """-> closure returning the LIST of values of a call / vararg expression (a single value for anything else)"""
        kind = node[0]
        if kind == 'vararg':
            if not fs.is_vararg:
                raise LuaError("%s:%d: cannot use '...' outside a vararg function" % (self.chunk, node[1]))
            return lambda fr: list(fr.va)
        if kind == 'call':
            fn, args, line, chunk = self.expr(node[1], fs), self.exprlist(node[2], fs), node[3], self.chunk
            desc = _describe(node[1])
            what = desc[2:-1] if desc else '?'

            def ev_call(fr):
                f = fn(fr)
                a = args(fr)
                try:
                    return call(f, a)
                except LuaError as e:
                    if not e.tb and isinstance(e.value, str) and e.value.startswith('attempt to call'):
                        e.value = '%s:%d: %s%s' % (chunk, line, e.value, desc)
                    if e.level is not None:
                        e.level -= 1
                        if e.level == 0:
                            e.value, e.level = '%s:%d: %s' % (chunk, line, e.value), None
                    e.tb.append((chunk, line, what))
                    raise
                except RecursionError:
                    raise LuaError('%s:%d: stack overflow' % (chunk, line))
            return ev_call
        if kind == 'method':
            obj, name, args, line, chunk = self.expr(node[1], fs), node[2], self.exprlist(node[3], fs), node[4], self.chunk
            desc = _describe(node[1])

            def ev_method(fr):
                o = obj(fr)
                try:
                    f = index(o, name)
                except LuaError as e:
                    if not e.tb and isinstance(e.value, str):
                        e.value = '%s:%d: %s%s' % (chunk, line, e.value, desc)
                    e.tb.append((chunk, line, 'method ' + name))
                    raise
                a = args(fr)
                a.insert(0, o)
                try:
                    return call(f, a)
                except LuaError as e:
                    if not e.tb and isinstance(e.value, str) and e.value.startswith('attempt to call'):
                        e.value = "%s:%d: %s (method '%s')" % (chunk, line, e.value, name)
                    if e.level is not None:
                        e.level -= 1
                        if e.level == 0:
                            e.value, e.level = '%s:%d: %s' % (chunk, line, e.value), None
                    e.tb.append((chunk, line, 'method ' + name))
                    raise
                except RecursionError:
                    raise LuaError('%s:%d: stack overflow' % (chunk, line))
            return ev_method
        if kind == 'paren':
            e = self.expr(node[1], fs)
            return lambda fr: [e(fr)]
        e = self.expr(node, fs)
        return lambda fr: [e(fr)]

    def exprlist(self, nodes, fs):
        """-> closure returning a fresh list: every expression truncated to one value except a trailing call / vararg"""
        if not nodes:
            return lambda fr: []
        last = nodes[-1]
        singles = [self.expr(n, fs) for n in nodes[:-1]]
        if last[0] in ('call', 'method', 'vararg'):
            m = self.multi(last, fs)
            if not singles:
                return lambda fr: list(m(fr))

            def ev_list_multi(fr):
                out = [f(fr) for f in singles]
                out.extend(m(fr))
                return out
            return ev_list_multi
        singles.append(self.expr(last, fs))
        if len(singles) == 1:
            f0 = singles[0]
            return lambda fr: [f0(fr)]
        if len(singles) == 2:
            f0, f1 = singles
            return lambda fr: [f0(fr), f1(fr)]
        return lambda fr: [f(fr) for f in singles]

    def function(self, node, fs):
        _, params, is_vararg, body, name, line = node
        sub = FuncState(fs)
        sub.is_vararg = is_vararg
        for p in params:
            sub.declare(p)
        code = self.block(body, sub, new_scope=False)
        proto = Proto()
        proto.name, proto.nparams, proto.is_vararg, proto.body = name, len(params), is_vararg, code
        proto.upvals, proto.chunk, proto.line = sub.upvals, self.chunk, line
        proto.nslots = sub.nslots           # final only now: read at call time through the proto
        vm, ups = self.vm, sub.upvals

        def ev_func(fr):
            return LuaFunction(proto, [fr.s[i] if from_local else fr.u[i] for from_local, i in ups], vm)
        return ev_func

    # ---- statements
    def block(self, stats, fs, new_scope=True):
        if new_scope:
            fs.blocks.append({})
        code = [self.stat(s, fs) for s in stats]
        if new_scope:
            fs.blocks.pop()
        if not code:
            return lambda fr: None
        if len(code) == 1:
            return code[0]

        def ex_block(fr):
            for c in code:
                r = c(fr)
                if r is not None:
                    return r
            return None
        return ex_block

    def assign_target(self, node, fs):
        """-> closure (fr, value)"""
        if node[0] == 'name':
            r = fs.resolve(node[1])
            if r[0] == 'local':
                i = r[1]

                def set_local(fr, v):
                    fr.s[i][0] = v
                return set_local
            if r[0] == 'up':
                i = r[1]

                def set_up(fr, v):
                    fr.u[i][0] = v
                return set_up
            G, name = self.G, node[1]

            def set_global(fr, v):
                G.set(name, v)
            return set_global
        obj, key, line, chunk = self.expr(node[1], fs), self.expr(node[2], fs), node[3], self.chunk
        desc = _describe(node[1])

        def set_index(fr, v):
            o = obj(fr)
            try:
                setindex(o, key(fr), v)
            except LuaError as e:
                if not e.tb:
                    if isinstance(e.value, str):
                        e.value = '%s:%d: %s%s' % (chunk, line, e.value, desc)
                    e.tb.append((chunk, line, 'newindex'))
                raise
        return set_index

    def stat(self, node, fs):
        kind = node[0]
        if kind == 'local':
            names, exprs = node[1], node[2]
            if len(names) == 1 and len(exprs) == 1:
                e = self.expr(exprs[0], fs)
                i = fs.declare(names[0])

                def ex_local1(fr):
                    fr.s[i] = [e(fr)]
                return ex_local1
            el = self.exprlist(exprs, fs)
            idx = [fs.declare(n) for n in names]       # declared after the initialisers are compiled
            nn = len(idx)

            def ex_local(fr):
                vals = el(fr)
                s = fr.s
                nv = len(vals)
                for j in range(nn):
                    s[idx[j]] = [vals[j] if j < nv else None]
            return ex_local
        if kind == 'assign':
            targets, exprs = node[1], node[2]
            if len(targets) == 1 and len(exprs) == 1:
                t, e = self.assign_target(targets[0], fs), self.expr(exprs[0], fs)

                def ex_assign1(fr):
                    t(fr, e(fr))
                return ex_assign1
            ts, el = [self.assign_target(t, fs) for t in targets], self.exprlist(exprs, fs)

            def ex_assign(fr):
                vals = el(fr)
                nv = len(vals)
                for j, t in enumerate(ts):
                    t(fr, vals[j] if j < nv else None)
            return ex_assign
        if kind == 'callstat':
            m = self.multi(node[1], fs)

            def ex_call(fr):
                m(fr)
            return ex_call
        if kind == 'do':
            return self.block(node[1], fs)
        if kind == 'return':
            exprs = node[1]
            if len(exprs) == 1 and exprs[0][0] not in ('call', 'method', 'vararg'):
                e = self.expr(exprs[0], fs)
                return lambda fr: [e(fr)]
            el = self.exprlist(exprs, fs)
            return lambda fr: el(fr)
        if kind == 'break':
            return lambda fr: BREAK
        if kind == 'if':
            clauses = [(self.expr(c, fs), self.block(b, fs)) for c, b in node[1]]
            orelse = self.block(node[2], fs) if node[2] is not None else None
            if len(clauses) == 1:
                c0, b0 = clauses[0]

                def ex_if1(fr):
                    v = c0(fr)
                    if v is not None and v is not False:
                        return b0(fr)
                    if orelse is not None:
                        return orelse(fr)
                return ex_if1

            def ex_if(fr):
                for c, b in clauses:
                    v = c(fr)
                    if v is not None and v is not False:
                        return b(fr)
                if orelse is not None:
                    return orelse(fr)
            return ex_if
        if kind == 'while':
            c, b = self.expr(node[1], fs), self.block(node[2], fs)

            def ex_while(fr):
                while True:
                    v = c(fr)
                    if v is None or v is False:
                        return None
                    r = b(fr)
                    if r is not None:
                        if r is BREAK:
                            return None
                        return r
            return ex_while
        if kind == 'repeat':
            fs.blocks.append({})
            b = self.block(node[1], fs, new_scope=False)
            c = self.expr(node[2], fs)                 # the condition sees the body's locals
            fs.blocks.pop()

            def ex_repeat(fr):
                while True:
                    r = b(fr)
                    if r is not None:
                        if r is BREAK:
                            return None
                        return r
                    v = c(fr)
                    if v is not None and v is not False:
                        return None
            return ex_repeat
        if kind == 'fornum':
            _, var, start, stop, step, body, line = node
            e0, e1 = self.expr(start, fs), self.expr(stop, fs)
            e2 = self.expr(step, fs) if step is not None else None
            fs.blocks.append({})
            i = fs.declare(var)
            b = self.block(body, fs)
            fs.blocks.pop()
            chunk = self.chunk

            def ex_fornum(fr):
                a, z = e0(fr), e1(fr)
                st = e2(fr) if e2 is not None else 1
                for what, v in (('initial', a), ('limit', z), ('step', st)):
                    if v.__class__ not in (int, float):
                        n = tonumber(v) if v.__class__ is str else None
                        if n is None:
                            raise LuaError("%s:%d: 'for' %s value must be a number" % (chunk, line, what), [(chunk, line, 'for')])
                a = a if a.__class__ in (int, float) else tonumber(a)
                z = z if z.__class__ in (int, float) else tonumber(z)
                st = st if st.__class__ in (int, float) else tonumber(st)
                s = fr.s
                if st > 0:
                    while a <= z:
                        s[i] = [a]
                        r = b(fr)
                        if r is not None:
                            if r is BREAK:
                                return None
                            return r
                        a += st
                elif st < 0:
                    while a >= z:
                        s[i] = [a]
                        r = b(fr)
                        if r is not None:
                            if r is BREAK:
                                return None
                            return r
                        a += st
                return None
            return ex_fornum
        if kind == 'forin':
            _, names, exprs, body, line = node
            el = self.exprlist(exprs, fs)
            fs.blocks.append({})
            idx = [fs.declare(n) for n in names]
            b = self.block(body, fs)
            fs.blocks.pop()
            nn, chunk = len(idx), self.chunk

            def ex_forin(fr):
                vals = el(fr)
                f = vals[0] if vals else None
                st = vals[1] if len(vals) > 1 else None
                ctl = vals[2] if len(vals) > 2 else None
                s = fr.s
                while True:
                    try:
                        rs = call(f, [st, ctl])
                    except LuaError as e:
                        e.tb.append((chunk, line, 'for iterator'))
                        raise
                    ctl = rs[0] if rs else None
                    if ctl is None:
                        return None
                    nr = len(rs)
                    for j in range(nn):
                        s[idx[j]] = [rs[j] if j < nr else None]
                    r = b(fr)
                    if r is not None:
                        if r is BREAK:
                            return None
                        return r
            return ex_forin
        if kind == 'localfunc':
            i = fs.declare(node[1])                    # visible inside its own body
            f = self.function(node[2], fs)

            def ex_localfunc(fr):
                cell = [None]
                fr.s[i] = cell
                cell[0] = f(fr)
            return ex_localfunc
        raise LuaError('compiler: unknown statement %r' % (kind,))


# ------------------------------------------------------------------------------------------------------------- Lua patterns
_CLASS = {'a': 'A-Za-z', 'd': '0-9', 'l': 'a-z', 's': r' \t\n\r\f\v', 'u': 'A-Z', 'w': 'A-Za-z0-9', 'x': '0-9A-Fa-f',
          'p': r'!-/:-@\[-`{-~', 'c': r'\x00-\x1f\x7f'}


def _class_re(ch, in_set):
    low = ch.lower()
    if low in _CLASS:
        body = _CLASS[low]
        if ch.islower():
            return body if in_set else '[' + body + ']'
        if in_set:
            raise LuaError('luavm: complemented class %%%s inside a set is not supported' % ch)
        return '[^' + body + ']'
    return re.escape(ch)


_PATCACHE = {}


def pattern_to_re(p):
    """Lua pattern -> compiled Python regex (no %b, %f; position captures '()' unsupported)"""
    if p in _PATCACHE:
        return _PATCACHE[p]
    out, i, n = [], 0, len(p)
    if p.startswith('^'):
        out.append(r'\A')
        i = 1
    while i < n:
        ch = p[i]
        if ch == '%':
            i += 1
            if i >= n:
                raise LuaError('malformed pattern (ends with %)')
            if p[i] in 'bf':
                raise LuaError('luavm: %%%s in patterns is not supported' % p[i])
            if p[i].isdigit():
                out.append('(?:\\%s)' % p[i])
            else:
                out.append(_class_re(p[i], False))
            i += 1
        elif ch == '[':
            j = i + 1
            body = '['
            if j < n and p[j] == '^':
                body += '^'
                j += 1
            start = True
            while j < n and (p[j] != ']' or start):
                start = False
                if p[j] == '%':
                    j += 1
                    body += _class_re(p[j], True)
                elif p[j] in '[\\':
                    body += '\\' + p[j]
                else:
                    body += p[j]
                j += 1
            if j >= n:
                raise LuaError('malformed pattern (missing ])')
            out.append(body + ']')
            i = j + 1
        elif ch == '(':
            if i + 1 < n and p[i + 1] == ')':
                raise LuaError('luavm: position captures are not supported')
            out.append('(')
            i += 1
        elif ch == ')':
            out.append(')')
            i += 1
        elif ch == '.':
            out.append('(?s:.)')
            i += 1
        elif ch == '-':
            out.append('*?')
            i += 1
        elif ch in '*+?':
            out.append(ch)
            i += 1
        elif ch == '$' and i == n - 1:
            out.append(r'\Z')
            i += 1
        else:
            out.append(re.escape(ch))
            i += 1
    rx = re.compile(''.join(out))
    _PATCACHE[p] = rx
    return rx


def _captures(m, whole_if_none=True):
    if m.re.groups == 0:
        return [m.group(0)] if whole_if_none else []
    return [g for g in m.groups()]


def _str_arg(v, n, fname):
    if v.__class__ is str:
        return v
    if v.__class__ in (int, float):
        return fmt_number(v)
    raise LuaError("bad argument #%d to '%s' (string expected, got %s)" % (n, fname, 'no value' if v is None else type_name(v)))


def _str_find(s, pat, init=1, plain=None, *_):
    s, pat = _str_arg(s, 1, 'find'), _str_arg(pat, 2, 'find')
    init = int(init or 1)
    if init < 0:
        init = max(len(s) + init + 1, 1)
    elif init == 0:
        init = 1
    if truthy(plain) or not re.search(r'[\^$*+?.()\[\]%-]', pat):
        i = s.find(pat, init - 1)
        if i < 0:
            return None
        return (i + 1, i + len(pat))
    m = pattern_to_re(pat).search(s, init - 1)
    if not m:
        return None
    return tuple([m.start() + 1, m.end()] + _captures(m, False))


def _str_match(s, pat, init=1, *_):
    s, pat = _str_arg(s, 1, 'match'), _str_arg(pat, 2, 'match')
    init = int(init or 1)
    if init < 0:
        init = max(len(s) + init + 1, 1)
    m = pattern_to_re(pat).search(s, max(init - 1, 0))
    if not m:
        return None
    return tuple(_captures(m))


def _str_gmatch(s, pat, *_):
    s, pat = _str_arg(s, 1, 'gmatch'), _str_arg(pat, 2, 'gmatch')
    it = pattern_to_re(pat).finditer(s)

    def step(*_a):
        for m in it:
            return tuple(_captures(m))
        return None
    return step


def _str_gsub(s, pat, repl, max_n=None, *_):
    s, pat = _str_arg(s, 1, 'gsub'), _str_arg(pat, 2, 'gsub')
    rx = pattern_to_re(pat)
    count = [0]

    def sub(m):
        count[0] += 1
        caps = _captures(m)
        if repl.__class__ in (str, int, float):
            r = _str_arg(repl, 3, 'gsub')
            out, i = [], 0
            while i < len(r):
                if r[i] == '%' and i + 1 < len(r):
                    d = r[i + 1]
                    if d == '0':
                        out.append(m.group(0))
                    elif d.isdigit():
                        out.append(caps[int(d) - 1])
                    else:
                        out.append(d)
                    i += 2
                else:
                    out.append(r[i])
                    i += 1
            return ''.join(out)
        if repl.__class__ is LuaTable:
            v = index(repl, caps[0])
        else:
            v = first(call(repl, caps))
        if v is None or v is False:
            return m.group(0)
        return _str_arg(v, 3, 'gsub')
    res = rx.sub(sub, s, count=int(max_n) if max_n is not None else 0)
    return (res, count[0])


_FMT = re.compile(r'%([-+ #0]*)(\d+)?(?:\.(\d+))?([a-zA-Z%])')


def _str_format(fmt, *args):
    fmt = _str_arg(fmt, 1, 'format')
    it = iter(range(len(args)))

    def one(m):
        flags, width, prec, conv = m.group(1), m.group(2) or '', m.group(3), m.group(4)
        if conv == '%':
            return '%'
        try:
            k = next(it)
        except StopIteration:
            raise LuaError("bad argument #%d to 'format' (no value)" % (len(args) + 2))
        v = args[k]
        spec = '%' + flags + width + ('.' + prec if prec is not None else '')
        if conv in 'di':
            n = v if v.__class__ in (int, float) else tonumber(v) if v.__class__ is str else None
            if n is None:
                raise LuaError("bad argument #%d to 'format' (number expected, got %s)" % (k + 2, type_name(v)))
            return (spec + 'd') % int(n)
        if conv in 'uoxX':
            n = v if v.__class__ in (int, float) else tonumber(v) if v.__class__ is str else None
            if n is None:
                raise LuaError("bad argument #%d to 'format' (number expected, got %s)" % (k + 2, type_name(v)))
            return (spec + ('d' if conv == 'u' else conv)) % int(n)
        if conv in 'eEfgG':
            n = v if v.__class__ in (int, float) else tonumber(v) if v.__class__ is str else None
            if n is None:
                raise LuaError("bad argument #%d to 'format' (number expected, got %s)" % (k + 2, type_name(v)))
            return (spec + conv) % float(n)
        if conv == 'c':
            return chr(int(v))
        if conv == 's':
            return (spec + 's') % tostring(v)
        if conv == 'q':
            return '"' + tostring(v).replace('\\', '\\\\').replace('"', '\\"').replace('\n', '\\\n') + '"'
        raise LuaError("invalid option '%%%s' to 'format'" % conv)
    return _FMT.sub(one, fmt)


def _str_sub(s, i=1, j=-1, *_):
    s = _str_arg(s, 1, 'sub')
    n = len(s)
    i, j = int(i if i is not None else 1), int(j if j is not None else -1)
    if i < 0:
        i = max(n + i + 1, 1)
    elif i == 0:
        i = 1
    if j < 0:
        j = n + j + 1
    elif j > n:
        j = n
    return s[i - 1:j] if i <= j else ''


def _str_byte(s, i=1, j=None, *_):
    s = _str_arg(s, 1, 'byte')
    i = int(i or 1)
    j = int(j) if j is not None else i
    sub = _str_sub(s, i, j)
    return tuple(ord(c) for c in sub) if sub else None


def _str_rep(s, n, *_):
    return _str_arg(s, 1, 'rep') * max(int(n), 0)


STRING_LIB = LuaTable()
for _n, _f in (('format', _str_format), ('find', _str_find), ('match', _str_match), ('gmatch', _str_gmatch), ('gsub', _str_gsub),
               ('sub', _str_sub), ('byte', _str_byte), ('rep', _str_rep),
               ('len', lambda s, *_: len(_str_arg(s, 1, 'len'))), ('lower', lambda s, *_: _str_arg(s, 1, 'lower').lower()),
               ('upper', lambda s, *_: _str_arg(s, 1, 'upper').upper()), ('reverse', lambda s, *_: _str_arg(s, 1, 'reverse')[::-1]),
               ('char', lambda *a: ''.join(chr(int(x)) for x in a))):
    STRING_LIB.set(_n, _f)


# ----------------------------------------------------------------------------------------------------------------- the machine
class LuaFile(object):
    lua_type = 'userdata'

    def __init__(self, f):
        self.f = f

    def lua_index(self, k):
        return getattr(self, 'm_' + k, None)

    def lua_tostring(self):
        return 'file (0x%08x)' % (id(self) & 0xffffffff)

    def _read_one(self, fmt):
        if fmt.__class__ in (int, float):
            d = self.f.read(int(fmt))
            return d if d or fmt == 0 else None
        fmt = fmt.lstrip('*')
        if fmt.startswith('a'):
            return self.f.read()
        if fmt.startswith('l'):
            d = self.f.readline()
            if not d:
                return None
            return d[:-1] if d.endswith('\n') else d
        if fmt.startswith('n'):
            d = self.f.readline()
            return tonumber(d.strip())
        raise LuaError("bad argument #1 to 'read' (invalid format)")

    def m_read(self, _self, *fmts):
        if not fmts:
            fmts = ('*l',)
        return tuple(self._read_one(f) for f in fmts)

    def m_write(self, _self, *vals):
        for v in vals:
            self.f.write(_str_arg(v, 1, 'write'))
        return self

    def m_lines(self, _self):
        def step(*_a):
            return self._read_one('*l')
        return step

    def m_close(self, _self=None):
        self.f.close()
        return True

    def m_flush(self, _self=None):
        self.f.flush()

    def m_seek(self, _self, whence='cur', off=0):
        self.f.seek(int(off), {'set': 0, 'cur': 1, 'end': 2}[whence])
        return self.f.tell()


def _os_date(fmt='%c', t=None, *_):
    """os.date: '*t' / '!*t' give the broken-down time as a table (year, month, day, hour, min, sec, wday, yday, isdst)"""
    utc = fmt.startswith('!')
    if utc:
        fmt = fmt[1:]
    tm = (time.gmtime if utc else time.localtime)(t)
    if fmt == '*t':
        out = LuaTable()
        for k, v in (('year', tm.tm_year), ('month', tm.tm_mon), ('day', tm.tm_mday), ('hour', tm.tm_hour), ('min', tm.tm_min),
                     ('sec', tm.tm_sec), ('wday', (tm.tm_wday + 1) % 7 + 1), ('yday', tm.tm_yday), ('isdst', bool(tm.tm_isdst))):
            out.set(k, v)
        return out
    return time.strftime(fmt, tm)


class LuaVM(object):
    """One Lua state.  search = directories that `dofile` / `require` resolve relative paths in (first hit wins)."""

    def __init__(self, search=(), stdout=None):
        sys.setrecursionlimit(max(sys.getrecursionlimit(), 20000))
        self.search = list(search)
        self.stdout = stdout or sys.stdout
        self.globals = LuaTable()
        self.loaded = LuaTable()
        self.preload = LuaTable()
        self.make_int64 = lambda v, unsigned: (_ for _ in ()).throw(LuaError('luavm: 64-bit literals need the ffi module'))
        self.at_close = []
        self._install_base()

    # ---- loading
    def find_file(self, path):
        if os.path.isabs(path):
            return path if os.path.exists(path) else None
        for d in self.search:
            p = os.path.join(d, path)
            if os.path.exists(p):
                return p
        return None

    def load(self, src, chunkname='=(load)'):
        ast = parse(src, chunkname)
        fs = FuncState(None)
        fs.is_vararg = True
        comp = Compiler(self, chunkname)
        code = comp.block(ast, fs, new_scope=False)
        proto = Proto()
        proto.name, proto.nparams, proto.is_vararg, proto.body = 'main chunk', 0, True, code
        proto.nslots, proto.upvals, proto.chunk, proto.line = fs.nslots, [], chunkname, 0
        return LuaFunction(proto, [], self)

    def loadfile(self, path):
        p = self.find_file(path)
        if p is None:
            raise LuaError('cannot open %s' % path)
        with open(p, 'rb') as f:
            src = f.read().decode('latin-1')
        return self.load(src, os.path.relpath(p, self.search[0]) if self.search and p.startswith(self.search[0]) else p)

    def dofile(self, path, *args):
        return first(call(self.loadfile(path), list(args)))

    def dostring(self, src, *args, **kw):
        return call(self.load(src, kw.get('name', '=(string)')), list(args))

    def require(self, name):
        v = self.loaded.get(name)
        if v is not None:
            return v
        loader = self.preload.get(name)
        if loader is None:
            rel = name.replace('.', '/')
            for cand in (rel + '.lua', rel + '/init.lua'):
                if self.find_file(cand):
                    loader = self.loadfile(cand)
                    break
        if loader is None:
            raise LuaError("module '%s' not found" % name)
        v = first(call(loader, [name]))
        if v is None:
            v = self.loaded.get(name)
            if v is None:
                v = True
        self.loaded.set(name, v)
        return v

    def call(self, f, *args):
        return call(f, list(args))

    def close(self):
        """run the pending ffi.gc finalisers (lua_close)"""
        for f in reversed(self.at_close):
            f()
        self.at_close = []

    # ---- base library
    def _install_base(self):
        G = self.globals
        vm = self

        def lua_print(*args):
            vm.stdout.write('\t'.join(tostring(a) for a in args).encode('latin-1', 'replace').decode('utf-8', 'replace') + '\n')

        def lua_error(msg=None, level=1, *_):
            if msg.__class__ is str and level and level > 0:
                raise LuaError(msg, [], int(level))
            raise LuaError(msg, [])

        def lua_assert(*args):
            if not args or not truthy(args[0]):
                raise LuaError(args[1] if len(args) > 1 else 'assertion failed!')
            return tuple(args)

        def lua_pcall(f=None, *args):
            try:
                return tuple([True] + call(f, list(args)))
            except LuaError as e:
                return (False, e.value)
            except RecursionError:
                return (False, 'stack overflow')

        def lua_xpcall(f, handler, *args):
            try:
                return tuple([True] + call(f, list(args)))
            except LuaError as e:
                return tuple([False] + call(handler, [e.value]))

        def lua_select(n, *args):
            if n == '#':
                return len(args)
            n = int(n)
            if n < 0:
                n = len(args) + n + 1
            if n < 1:
                raise LuaError("bad argument #1 to 'select' (index out of range)")
            return tuple(args[n - 1:])

        def lua_unpack(t, i=1, j=None, *_):
            if t.__class__ is not LuaTable:
                raise LuaError("bad argument #1 to 'unpack' (table expected, got %s)" % type_name(t))
            j = t.length() if j is None else j
            return tuple(t.get(k) for k in range(int(i), int(j) + 1))

        def lua_next(t, k=None, *_):
            if t.__class__ is not LuaTable:
                raise LuaError("bad argument #1 to 'next' (table expected, got %s)" % type_name(t))
            nk, nv = t.next(k)
            return (nk, nv) if nk is not None else None

        def lua_pairs(t=None, *_):
            if t.__class__ is not LuaTable:
                p = getattr(t, 'lua_pairs', None)
                if p is not None:
                    return p()
                raise LuaError("bad argument #1 to 'pairs' (table expected, got %s)" % type_name(t))
            keys = t.keys()
            pos = [0]

            def step(_t, _k):
                while pos[0] < len(keys):
                    k = keys[pos[0]]
                    pos[0] += 1
                    v = t.get(k)
                    if v is not None:
                        return (k, v)
                return None
            return (step, t, None)

        def lua_ipairs(t=None, *_):
            if t.__class__ is not LuaTable and not hasattr(t, 'lua_index'):
                raise LuaError("bad argument #1 to 'ipairs' (table expected, got %s)" % type_name(t))

            def step(tt, i):
                i += 1
                v = index(tt, i)
                if v is None:
                    return None
                return (i, v)
            return (step, t, 0)

        def lua_setmetatable(t, m=None, *_):
            if t.__class__ is not LuaTable:
                raise LuaError("bad argument #1 to 'setmetatable' (table expected, got %s)" % type_name(t))
            if m is not None and m.__class__ is not LuaTable:
                raise LuaError("bad argument #2 to 'setmetatable' (nil or table expected)")
            if t.meta is not None and t.meta.get('__metatable') is not None:
                raise LuaError('cannot change a protected metatable')
            t.meta = m
            return t

        def lua_getmetatable(t=None, *_):
            if t.__class__ is LuaTable:
                if t.meta is None:
                    return None
                p = t.meta.get('__metatable')
                return p if p is not None else t.meta
            if t.__class__ is str:
                m = LuaTable()
                m.set('__index', STRING_LIB)
                return m
            g = getattr(t, 'lua_getmetatable', None)
            return g() if g is not None else None

        def lua_rawget(t, k, *_):
            return t.get(k)

        def lua_rawset(t, k, v=None, *_):
            t.set(k, v)
            return t

        def lua_type(*args):
            if not args:
                raise LuaError("bad argument #1 to 'type' (value expected)")
            return type_name(args[0])

        def lua_tostring(*args):
            if not args:
                raise LuaError("bad argument #1 to 'tostring' (value expected)")
            return tostring(args[0])

        def lua_loadstring(src, name=None, *_):
            try:
                return vm.load(src, name or '=(loadstring)')
            except Exception as e:                 # syntax errors are returned, not raised
                return (None, str(e))

        def lua_dofile(path=None, *_):
            return tuple(call(vm.loadfile(path), []))

        def lua_loadfile(path=None, *_):
            try:
                return vm.loadfile(path)
            except Exception as e:
                return (None, str(e))

        for name, f in (('print', lua_print), ('error', lua_error), ('assert', lua_assert), ('pcall', lua_pcall),
                        ('xpcall', lua_xpcall), ('select', lua_select), ('unpack', lua_unpack), ('next', lua_next),
                        ('pairs', lua_pairs), ('ipairs', lua_ipairs), ('setmetatable', lua_setmetatable),
                        ('getmetatable', lua_getmetatable), ('rawget', lua_rawget), ('rawset', lua_rawset),
                        ('rawequal', lambda a, b, *_: a is b or (a.__class__ in (int, float, str) and lua_eq(a, b))),
                        ('type', lua_type), ('tostring', lua_tostring), ('tonumber', lambda v=None, b=None, *_: tonumber(v, b)),
                        ('loadstring', lua_loadstring), ('dofile', lua_dofile), ('loadfile', lua_loadfile),
                        ('require', lambda name, *_: vm.require(name)),
                        ('collectgarbage', lambda opt='collect', *_: 0 if opt == 'count' else (__import__('gc').collect() and 0))):
            G.set(name, f)
        G.set('_G', G)
        G.set('_VERSION', 'Lua 5.1')
        G.set('string', STRING_LIB)

        # table
        T = LuaTable()

        def t_insert(t, *args):
            if len(args) == 1:
                t.set(t.length() + 1, args[0])
            elif len(args) == 2:
                pos, v = int(args[0]), args[1]
                n = t.length()
                for k in range(n, pos - 1, -1):
                    t.set(k + 1, t.get(k))
                t.set(pos, v)
            else:
                raise LuaError("wrong number of arguments to 'insert'")

        def t_remove(t, pos=None, *_):
            n = t.length()
            if n == 0:
                return None
            pos = n if pos is None else int(pos)
            v = t.get(pos)
            for k in range(pos, n):
                t.set(k, t.get(k + 1))
            t.set(n, None)
            return v

        def t_concat(t, sep='', i=1, j=None, *_):
            j = t.length() if j is None else j
            parts = []
            for k in range(int(i), int(j) + 1):
                v = t.get(k)
                if v.__class__ not in (str, int, float):
                    raise LuaError("invalid value (at index %d) in table for 'concat'" % k)
                parts.append(v if v.__class__ is str else fmt_number(v))
            return (sep or '').join(parts)

        def t_sort(t, comp=None, *_):
            n = t.length()
            vals = [t.get(k) for k in range(1, n + 1)]
            if comp is None:
                lt = lua_lt
            else:
                def lt(a, b):
                    return truthy(first(call(comp, [a, b])))

            def cmp(a, b):
                if lt(a, b):
                    return -1
                if lt(b, a):
                    return 1
                return 0
            vals.sort(key=cmp_to_key(cmp))
            for k, v in enumerate(vals):
                t.set(k + 1, v)

        for name, f in (('insert', t_insert), ('remove', t_remove), ('concat', t_concat), ('sort', t_sort),
                        ('getn', lambda t, *_: t.length()), ('maxn', lambda t, *_: max([k for k in t.keys() if k.__class__ in (int, float)] or [0]))):
            T.set(name, f)
        G.set('table', T)

        # math
        M = LuaTable()

        def m_floor(x, *_):
            r = math.floor(x)
            return r

        def m_random(a=None, b=None, *_):
            import random
            if a is None:
                return vm_random.random()
            if b is None:
                return vm_random.randint(1, int(a))
            return vm_random.randint(int(a), int(b))
        import random as _random
        vm_random = _random.Random(0)
        for name, f in (('floor', m_floor), ('ceil', lambda x, *_: math.ceil(x)), ('sqrt', lambda x, *_: math.sqrt(x) if x >= 0 else float('nan')),
                        ('exp', lambda x, *_: math.exp(x)), ('log', lambda x, *_: math.log(x) if x > 0 else (float('-inf') if x == 0 else float('nan'))),
                        ('log10', lambda x, *_: math.log10(x)), ('abs', lambda x, *_: abs(x)),
                        ('max', lambda *a: max(a)), ('min', lambda *a: min(a)), ('pow', lambda a, b, *_: arith('^', a, b)),
                        ('sin', lambda x, *_: math.sin(x)), ('cos', lambda x, *_: math.cos(x)), ('tan', lambda x, *_: math.tan(x)),
                        ('tanh', lambda x, *_: math.tanh(x)), ('fmod', lambda a, b, *_: math.fmod(a, b)),
                        ('modf', lambda x, *_: (float(int(x)), x - int(x))), ('random', m_random),
                        ('randomseed', lambda s=0, *_: vm_random.seed(int(s)))):
            M.set(name, f)
        M.set('pi', math.pi)
        M.set('huge', float('inf'))
        G.set('math', M)

        # os / io
        O = LuaTable()
        O.set('getenv', lambda k, *_: os.environ.get(k))
        O.set('time', lambda *_: int(time.time()))
        O.set('clock', lambda *_: time.process_time())
        O.set('date', _os_date)
        O.set('exit', lambda code=0, *_: (_ for _ in ()).throw(SystemExit(int(code if code.__class__ in (int, float) else 0))))
        def os_remove(p, *_):
            # the VM also runs the REFERENCE's scripts (untrusted): deletions are confined to the temporary directory
            import tempfile
            rp, root = os.path.realpath(str(p)), os.path.realpath(tempfile.gettempdir())
            if not rp.startswith(root + os.sep):
                return (None, '%s: os.remove outside %s is disabled in the test VM' % (p, root))
            os.remove(rp)
            return True
        O.set('remove', os_remove)
        G.set('os', O)
        IO = LuaTable()

        def io_open(path, mode='r', *_):
            try:
                p = path
                if 'r' in mode and not os.path.isabs(path):
                    p = vm.find_file(path) or path
                if 'r' not in mode or '+' in mode:      # writes: temporary directory only (see os.remove)
                    import tempfile
                    rp, root = os.path.realpath(str(p)), os.path.realpath(tempfile.gettempdir())
                    if not rp.startswith(root + os.sep):
                        return (None, '%s: writing outside %s is disabled in the test VM' % (path, root), 13)
                return LuaFile(open(p, mode.replace('b', ''), encoding='latin-1', newline=''))
            except IOError as e:
                return (None, '%s: %s' % (path, e.strerror), e.errno)

        def io_write(*vals):
            for v in vals:
                vm.stdout.write(_str_arg(v, 1, 'write').encode('latin-1', 'replace').decode('utf-8', 'replace'))
        IO.set('open', io_open)
        IO.set('write', io_write)
        IO.set('stdout', LuaFile(self.stdout))
        IO.set('stderr', LuaFile(sys.stderr))

        def io_lines(path, *_):
            f = io_open(path)
            if f.__class__ is tuple:
                raise LuaError(f[1])
            return f.m_lines(f)
        IO.set('lines', io_lines)
        G.set('io', IO)
        P = LuaTable()
        P.set('loaded', self.loaded)
        P.set('preload', self.preload)
        P.set('path', './?.lua')
        G.set('package', P)
        self.loaded.set('string', STRING_LIB)
        self.loaded.set('table', T)
        self.loaded.set('math', M)
        self.loaded.set('os', O)
        self.loaded.set('io', IO)
        self.loaded.set('_G', G)
