"""A Lua 5.1 syntax checker + undeclared-name lint, in Python -- test infrastructure for lua/*.lua.

No Lua interpreter exists in the build container or on the GPU box, so the hand-written Lua host (lua/model.lua, model_ops.lua,
vdnn.lua, the plug-in files) can never be executed here.  This module at least PARSES it: a complete recursive-descent parser of
the Lua 5.1 grammar (statements, expressions with the reference manual's precedences, table constructors, function bodies, method
calls, long strings / comments, numeric literals incl. LuaJIT's LL / ULL suffixes) that raises LuaSyntaxError with line numbers,
and a scope walk that reports every name that is read or called without being a local, a parameter, a loop variable or one of
the known globals -- the misspelt-local class of bug that otherwise only shows at run time.
Used by tests/test_lua_surface_cpu.py."""
import re

KEYWORDS = {'and', 'break', 'do', 'else', 'elseif', 'end', 'false', 'for', 'function', 'if', 'in', 'local', 'nil', 'not', 'or',
            'repeat', 'return', 'then', 'true', 'until', 'while'}


class LuaSyntaxError(Exception):
    pass


def tokenize(src):
    """-> list of (kind, text, line); kinds: name, kw, num, str, op, eof"""
    toks, i, n, line = [], 0, len(src), 1
    ops3, ops2 = ('...',), ('==', '~=', '<=', '>=', '..')
    while i < n:
        ch = src[i]
        if ch == '\n':
            line += 1
            i += 1
        elif ch in ' \t\r':
            i += 1
        elif src.startswith('--', i):
            m = re.match(r'--\[(=*)\[', src[i:])
            if m:
                close = ']' + m.group(1) + ']'
                j = src.find(close, i + len(m.group(0)))
                if j < 0:
                    raise LuaSyntaxError('line %d: unfinished long comment' % line)
                line += src.count('\n', i, j)
                i = j + len(close)
            else:
                j = src.find('\n', i)
                i = n if j < 0 else j
        elif ch == '[' and re.match(r'\[(=*)\[', src[i:]):
            m = re.match(r'\[(=*)\[', src[i:])
            close = ']' + m.group(1) + ']'
            j = src.find(close, i + len(m.group(0)))
            if j < 0:
                raise LuaSyntaxError('line %d: unfinished long string' % line)
            toks.append(('str', src[i:j + len(close)], line))
            line += src.count('\n', i, j)
            i = j + len(close)
        elif ch in '"\'':
            j = i + 1
            while True:
                if j >= n or src[j] == '\n':
                    raise LuaSyntaxError('line %d: unfinished string' % line)
                if src[j] == '\\':
                    j += 2
                    continue
                if src[j] == ch:
                    break
                j += 1
            toks.append(('str', src[i:j + 1], line))
            i = j + 1
        elif ch.isdigit() or (ch == '.' and i + 1 < n and src[i + 1].isdigit()):
            m = re.match(r'0[xX][0-9a-fA-F]+(?:ULL|LL|ull|ll)?|(?:\d+\.?\d*|\.\d+)(?:[eE][+-]?\d+)?(?:ULL|LL|ull|ll)?', src[i:])
            toks.append(('num', m.group(0), line))
            i += len(m.group(0))
            if i < n and (src[i].isalnum() or src[i] == '_'):
                raise LuaSyntaxError('line %d: malformed number near %r' % (line, src[i - 3:i + 3]))
        elif ch.isalpha() or ch == '_':
            m = re.match(r'[A-Za-z_][A-Za-z_0-9]*', src[i:])
            w = m.group(0)
            toks.append(('kw' if w in KEYWORDS else 'name', w, line))
            i += len(w)
        else:
            for ops in (ops3, ops2):
                hit = next((o for o in ops if src.startswith(o, i)), None)
                if hit:
                    break
            if hit:
                toks.append(('op', hit, line))
                i += len(hit)
            elif ch in '+-*/%^#<>=(){}[];:,.':
                toks.append(('op', ch, line))
                i += 1
            else:
                raise LuaSyntaxError('line %d: unexpected character %r' % (line, ch))
    toks.append(('eof', '<eof>', line))
    return toks


# binary operator -> (left priority, right priority), Lua 5.1 manual 2.5.6
BINPRI = {'or': (1, 1), 'and': (2, 2), '<': (3, 3), '>': (3, 3), '<=': (3, 3), '>=': (3, 3), '~=': (3, 3), '==': (3, 3),
          '..': (5, 4), '+': (6, 6), '-': (6, 6), '*': (7, 7), '/': (7, 7), '%': (7, 7), '^': (10, 9)}
UNARY_PRI = 8


class Parser(object):
    def __init__(self, src, name='<lua>', known_globals=()):
        self.t = tokenize(src)
        self.p = 0
        self.name = name
        self.scopes = [set()]
        self.globals = set(known_globals)
        self.undeclared = []          # (name, line) read or called without a declaration
        self.assigned_globals = set()  # names assigned at file level without `local` (they become globals)

    # ---- token helpers
    def peek(self, k=0):
        return self.t[min(self.p + k, len(self.t) - 1)]

    def err(self, what):
        kind, text, line = self.peek()
        raise LuaSyntaxError('%s:%d: %s near %r' % (self.name, line, what, text))

    def check(self, text):
        kind, tx, _ = self.peek()
        return kind in ('op', 'kw') and tx == text

    def accept(self, text):
        if self.check(text):
            self.p += 1
            return True
        return False

    def expect(self, text):
        if not self.accept(text):
            self.err('%r expected' % text)

    def expect_name(self):
        kind, tx, _ = self.peek()
        if kind != 'name':
            self.err('name expected')
        self.p += 1
        return tx

    # ---- scopes
    def declare(self, name):
        self.scopes[-1].add(name)

    def is_declared(self, name):
        return any(name in s for s in self.scopes)

    def use(self, name, line):
        if not self.is_declared(name) and name not in self.globals and name not in self.assigned_globals:
            self.undeclared.append((name, line))

    # ---- grammar
    def chunk(self):
        self.block()
        if self.peek()[0] != 'eof':
            self.err('unexpected token')

    def block_end(self):
        kind, tx, _ = self.peek()
        return kind == 'eof' or (kind == 'kw' and tx in ('end', 'else', 'elseif', 'until'))

    def block(self, scope=True):
        if scope:
            self.scopes.append(set())
        while not self.block_end():
            if self.check('return'):
                self.p += 1
                if not self.block_end() and not self.check(';'):
                    self.exprlist()
                self.accept(';')
                if not self.block_end():
                    self.err("'return' must be the last statement of a block")
                break
            if self.check('break'):
                self.p += 1
                self.accept(';')
                if not self.block_end():
                    self.err("'break' must be the last statement of a block")
                break
            self.statement()
            self.accept(';')
        if scope:
            self.scopes.pop()

    def statement(self):
        kind, tx, line = self.peek()
        if kind == 'kw':
            if tx == 'if':
                self.p += 1
                self.expr(); self.expect('then'); self.block()
                while self.accept('elseif'):
                    self.expr(); self.expect('then'); self.block()
                if self.accept('else'):
                    self.block()
                self.expect('end')
                return
            if tx == 'while':
                self.p += 1
                self.expr(); self.expect('do'); self.block(); self.expect('end')
                return
            if tx == 'do':
                self.p += 1
                self.block(); self.expect('end')
                return
            if tx == 'for':
                self.p += 1
                names = [self.expect_name()]
                if self.accept('='):
                    self.expr(); self.expect(','); self.expr()
                    if self.accept(','):
                        self.expr()
                else:
                    while self.accept(','):
                        names.append(self.expect_name())
                    self.expect('in')
                    self.exprlist()
                self.expect('do')
                self.scopes.append(set(names))
                self.block()
                self.scopes.pop()
                self.expect('end')
                return
            if tx == 'repeat':
                self.p += 1
                self.scopes.append(set())
                self.block(scope=False)
                self.expect('until')
                self.expr()                      # the condition sees the block's locals
                self.scopes.pop()
                return
            if tx == 'function':
                self.p += 1
                first = self.expect_name()
                line0 = self.peek()[2]
                is_method, dotted = False, False
                while self.accept('.'):
                    self.expect_name(); dotted = True
                if self.accept(':'):
                    self.expect_name(); is_method = True; dotted = True
                if dotted:
                    self.use(first, line0)
                elif not self.is_declared(first):
                    self.assigned_globals.add(first)
                self.funcbody(is_method)
                return
            if tx == 'local':
                self.p += 1
                if self.accept('function'):
                    name = self.expect_name()
                    self.declare(name)           # visible inside its own body (recursion)
                    self.funcbody(False)
                    return
                names = [self.expect_name()]
                while self.accept(','):
                    names.append(self.expect_name())
                if self.accept('='):
                    self.exprlist()
                for nm in names:                 # declared AFTER the initialisers are evaluated
                    self.declare(nm)
                return
            self.err('unexpected keyword')
        # exprstat: assignment or call
        kind_of, target = self.suffixedexp(for_assign=True)
        if self.check('=') or self.check(','):
            targets = [(kind_of, target)]
            while self.accept(','):
                targets.append(self.suffixedexp(for_assign=True))
            self.expect('=')
            self.exprlist()
            for k, nm in targets:
                if k == 'call':
                    self.err('cannot assign to a call')
                if k == 'name' and not self.is_declared(nm):
                    self.assigned_globals.add(nm)
        elif kind_of != 'call':
            self.err('syntax error (statement is neither an assignment nor a call)')

    def funcbody(self, is_method):
        self.expect('(')
        params = set(['self']) if is_method else set()
        if not self.check(')'):
            while True:
                if self.accept('...'):
                    params.add('...')
                    break
                params.add(self.expect_name())
                if not self.accept(','):
                    break
        self.expect(')')
        self.scopes.append(params)
        self.block()
        self.scopes.pop()
        self.expect('end')

    def exprlist(self):
        self.expr()
        while self.accept(','):
            self.expr()

    def primaryexp(self, for_assign):
        kind, tx, line = self.peek()
        if kind == 'name':
            self.p += 1
            return 'name', tx, line
        if self.accept('('):
            self.expr()
            self.expect(')')
            return 'paren', None, line
        self.err('unexpected symbol')

    def suffixedexp(self, for_assign=False):
        kind_of, nm, line = self.primaryexp(for_assign)
        first_name, first_line, plain = nm, line, kind_of == 'name'
        while True:
            kind, tx, _ = self.peek()
            if self.check('.'):
                self.p += 1; self.expect_name(); kind_of = 'index'
            elif self.check('['):
                self.p += 1; self.expr(); self.expect(']'); kind_of = 'index'
            elif self.check(':'):
                self.p += 1; self.expect_name(); self.callargs(); kind_of = 'call'
            elif self.check('(') or self.check('{') or kind == 'str':
                self.callargs(); kind_of = 'call'
            else:
                break
            if plain:                         # the base name is READ as soon as it is indexed or called
                self.use(first_name, first_line)
                plain = False
        if plain and not for_assign:
            self.use(first_name, first_line)
        elif plain and for_assign and not (self.check('=') or self.check(',')):
            self.use(first_name, first_line)
        return kind_of, first_name if kind_of == 'name' else None

    def callargs(self):
        kind, tx, _ = self.peek()
        if kind == 'str':
            self.p += 1
        elif self.check('{'):
            self.table()
        else:
            self.expect('(')
            if not self.check(')'):
                self.exprlist()
            self.expect(')')

    def table(self):
        self.expect('{')
        while not self.check('}'):
            kind, tx, _ = self.peek()
            if self.check('['):
                self.p += 1; self.expr(); self.expect(']'); self.expect('='); self.expr()
            elif kind == 'name' and self.peek(1)[1] == '=' and self.peek(1)[0] == 'op':
                self.p += 2; self.expr()
            else:
                self.expr()
            if not (self.accept(',') or self.accept(';')):
                break
        self.expect('}')

    def simpleexp(self):
        kind, tx, line = self.peek()
        if kind in ('num', 'str'):
            self.p += 1
            return
        if kind == 'kw' and tx in ('nil', 'true', 'false'):
            self.p += 1
            return
        if self.check('...'):
            self.p += 1
            if not self.is_declared('...'):
                self.err("cannot use '...' outside a vararg function")
            return
        if self.check('{'):
            self.table()
            return
        if self.check('function'):
            self.p += 1
            self.funcbody(False)
            return
        self.suffixedexp()

    def expr(self, limit=0):
        kind, tx, _ = self.peek()
        if (kind == 'kw' and tx == 'not') or (kind == 'op' and tx in ('-', '#')):
            self.p += 1
            self.expr(UNARY_PRI)
        else:
            self.simpleexp()
        while True:
            kind, tx, _ = self.peek()
            if kind not in ('op', 'kw') or tx not in BINPRI:
                break
            left, right = BINPRI[tx]
            if left <= limit:
                break
            self.p += 1
            self.expr(right)


LUA_GLOBALS = {'assert', 'collectgarbage', 'dofile', 'error', 'getmetatable', 'ipairs', 'io', 'math', 'next', 'os', 'pairs', 'pcall',
               'print', 'rawget', 'rawset', 'require', 'select', 'setmetatable', 'string', 'table', 'tonumber', 'tostring', 'type',
               'unpack', 'xpcall', 'package', 'arg', '_G', 'jit', 'bit', 'loadstring', 'coroutine', 'debug'}


def check(src, name='<lua>', extra_globals=()):
    """parse `src`; returns the list of (name, line) that are used without a declaration (empty = clean).  Raises LuaSyntaxError."""
    p = Parser(src, name, LUA_GLOBALS | set(extra_globals))
    p.chunk()
    return [(n, l) for n, l in p.undeclared if n not in p.assigned_globals]
