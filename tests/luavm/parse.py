"""Lua 5.1 source -> AST (nested tuples).  Grammar and operator precedences follow the Lua 5.1 reference manual (2.5.6, 8).

Expressions   ('nil',) ('true',) ('false',) ('num', v) ('str', s) ('vararg',) ('name', n, line)
              ('index', obj, key, line) ('call', fn, args, line) ('method', obj, name, args, line)
              ('func', params, is_vararg, body, name, line) ('bin', op, a, b, line) ('un', op, a, line)
              ('and', a, b) ('or', a, b) ('table', [(kind, key, value)], line) ('paren', e)
Statements    ('local', names, exprs, line) ('assign', targets, exprs, line) ('callstat', call, line) ('do', block)
              ('while', cond, block) ('repeat', block, cond) ('if', [(cond, block)], else_block)
              ('fornum', var, start, stop, step, block, line) ('forin', names, exprs, block, line)
              ('localfunc', name, func, line) ('return', exprs, line) ('break',)
Strings are Python str with one character per BYTE (latin-1), so binary data and `#s` behave like Lua's.
"""
import re


class LuaSyntaxError(Exception):
    pass


KEYWORDS = {'and', 'break', 'do', 'else', 'elseif', 'end', 'false', 'for', 'function', 'if', 'in', 'local', 'nil', 'not', 'or',
            'repeat', 'return', 'then', 'true', 'until', 'while'}
_NUM = re.compile(r'0[xX][0-9a-fA-F]+|(?:\d+\.?\d*|\.\d+)(?:[eE][+-]?\d+)?')
_NAME = re.compile(r'[A-Za-z_][A-Za-z_0-9]*')
_LONG = re.compile(r'\[(=*)\[')
_ESC = {'n': '\n', 't': '\t', 'r': '\r', 'a': '\a', 'b': '\b', 'f': '\f', 'v': '\v', '\\': '\\', '"': '"', "'": "'", '\n': '\n'}


def _unescape(body, line):
    out, i, n = [], 0, len(body)
    while i < n:
        ch = body[i]
        if ch != '\\':
            out.append(ch)
            i += 1
            continue
        i += 1
        c = body[i]
        if c in _ESC:
            out.append(_ESC[c])
            i += 1
        elif c.isdigit():
            j = i
            while j < n and j < i + 3 and body[j].isdigit():
                j += 1
            v = int(body[i:j])
            if v > 255:
                raise LuaSyntaxError('line %d: escape sequence too large' % line)
            out.append(chr(v))
            i = j
        else:
            raise LuaSyntaxError('line %d: invalid escape sequence \\%s' % (line, c))
    return ''.join(out)


def tokenize(src, name='<lua>'):
    """-> list of (kind, value, line); kinds: name, kw, num, str, op, eof"""
    toks, i, n, line = [], 0, len(src), 1
    if src.startswith('#'):                      # shebang line
        i = src.find('\n')
        i = n if i < 0 else i
    while i < n:
        ch = src[i]
        if ch == '\n':
            line += 1
            i += 1
        elif ch in ' \t\r':
            i += 1
        elif src.startswith('--', i):
            m = _LONG.match(src, i + 2)
            if m:
                close = ']' + m.group(1) + ']'
                j = src.find(close, m.end())
                if j < 0:
                    raise LuaSyntaxError('%s:%d: unfinished long comment' % (name, line))
                line += src.count('\n', i, j)
                i = j + len(close)
            else:
                j = src.find('\n', i)
                i = n if j < 0 else j
        elif ch == '[' and _LONG.match(src, i):
            m = _LONG.match(src, i)
            close = ']' + m.group(1) + ']'
            j = src.find(close, m.end())
            if j < 0:
                raise LuaSyntaxError('%s:%d: unfinished long string' % (name, line))
            body = src[m.end():j]
            if body.startswith('\r\n'):
                body = body[2:]
            elif body.startswith('\n'):
                body = body[1:]
            toks.append(('str', body, line))
            line += src.count('\n', i, j)
            i = j + len(close)
        elif ch in '"\'':
            j = i + 1
            while True:
                if j >= n or src[j] == '\n':
                    raise LuaSyntaxError('%s:%d: unfinished string' % (name, line))
                if src[j] == '\\':
                    if j + 1 < n and src[j + 1] == '\n':
                        line += 1
                    j += 2
                    continue
                if src[j] == ch:
                    break
                j += 1
            toks.append(('str', _unescape(src[i + 1:j], line), line))
            i = j + 1
        elif ch.isdigit() or (ch == '.' and i + 1 < n and src[i + 1].isdigit()):
            m = _NUM.match(src, i)
            text = m.group(0)
            i = m.end()
            suffix = re.match(r'ULL|LL|ull|ll', src[i:i + 3])
            if suffix:
                toks.append(('num64', (int(text, 0), suffix.group(0).upper() == 'ULL'), line))
                i += len(suffix.group(0))
            elif text[:2] in ('0x', '0X'):
                toks.append(('num', int(text, 16), line))
            else:
                v = float(text)
                toks.append(('num', int(v) if (v.is_integer() and abs(v) < 2 ** 53 and re.fullmatch(r'\d+', text)) else v, line))
            if i < n and (src[i].isalnum() or src[i] == '_'):
                raise LuaSyntaxError('%s:%d: malformed number near %r' % (name, line, src[i - 3:i + 3]))
        elif ch.isalpha() or ch == '_':
            m = _NAME.match(src, i)
            w = m.group(0)
            toks.append(('kw' if w in KEYWORDS else 'name', w, line))
            i = m.end()
        else:
            if src.startswith('...', i):
                toks.append(('op', '...', line))
                i += 3
            elif src[i:i + 2] in ('==', '~=', '<=', '>=', '..'):
                toks.append(('op', src[i:i + 2], line))
                i += 2
            elif ch in '+-*/%^#<>=(){}[];:,.':
                toks.append(('op', ch, line))
                i += 1
            else:
                raise LuaSyntaxError('%s:%d: unexpected character %r' % (name, line, ch))
    toks.append(('eof', '<eof>', line))
    return toks


# binary operator -> (left priority, right priority), Lua 5.1 manual 2.5.6
BINPRI = {'or': (1, 1), 'and': (2, 2), '<': (3, 3), '>': (3, 3), '<=': (3, 3), '>=': (3, 3), '~=': (3, 3), '==': (3, 3),
          '..': (5, 4), '+': (6, 6), '-': (6, 6), '*': (7, 7), '/': (7, 7), '%': (7, 7), '^': (10, 9)}
UNARY_PRI = 8


class Parser(object):
    def __init__(self, src, name='<lua>'):
        self.t = tokenize(src, name)
        self.p = 0
        self.name = name

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

    def line(self):
        return self.peek()[2]

    # ---- statements
    def chunk(self):
        body = self.block()
        if self.peek()[0] != 'eof':
            self.err('unexpected token')
        return body

    def block_end(self):
        kind, tx, _ = self.peek()
        return kind == 'eof' or (kind == 'kw' and tx in ('end', 'else', 'elseif', 'until'))

    def block(self):
        stats = []
        while not self.block_end():
            if self.check('return'):
                line = self.line()
                self.p += 1
                exprs = []
                if not self.block_end() and not self.check(';'):
                    exprs = self.exprlist()
                self.accept(';')
                if not self.block_end():
                    self.err("'return' must be the last statement of a block")
                stats.append(('return', exprs, line))
                break
            if self.check('break'):
                self.p += 1
                self.accept(';')
                if not self.block_end():
                    self.err("'break' must be the last statement of a block")
                stats.append(('break',))
                break
            stats.append(self.statement())
            self.accept(';')
        return stats

    def statement(self):
        kind, tx, line = self.peek()
        if kind == 'kw':
            if tx == 'if':
                self.p += 1
                clauses = []
                c = self.expr()
                self.expect('then')
                clauses.append((c, self.block()))
                orelse = None
                while True:
                    if self.accept('elseif'):
                        c = self.expr()
                        self.expect('then')
                        clauses.append((c, self.block()))
                    elif self.accept('else'):
                        orelse = self.block()
                        self.expect('end')
                        break
                    else:
                        self.expect('end')
                        break
                return ('if', clauses, orelse)
            if tx == 'while':
                self.p += 1
                c = self.expr()
                self.expect('do')
                b = self.block()
                self.expect('end')
                return ('while', c, b)
            if tx == 'do':
                self.p += 1
                b = self.block()
                self.expect('end')
                return ('do', b)
            if tx == 'for':
                self.p += 1
                names = [self.expect_name()]
                if self.accept('='):
                    start = self.expr()
                    self.expect(',')
                    stop = self.expr()
                    step = self.expr() if self.accept(',') else None
                    self.expect('do')
                    b = self.block()
                    self.expect('end')
                    return ('fornum', names[0], start, stop, step, b, line)
                while self.accept(','):
                    names.append(self.expect_name())
                self.expect('in')
                exprs = self.exprlist()
                self.expect('do')
                b = self.block()
                self.expect('end')
                return ('forin', names, exprs, b, line)
            if tx == 'repeat':
                self.p += 1
                b = self.block()
                self.expect('until')
                return ('repeat', b, self.expr())
            if tx == 'function':
                self.p += 1
                nline = self.line()
                full = self.expect_name()
                target = ('name', full, nline)
                is_method = False
                while self.check('.') or self.check(':'):
                    colon = self.check(':')
                    self.p += 1
                    key = self.expect_name()
                    full += (':' if colon else '.') + key
                    target = ('index', target, ('str', key), nline)
                    if colon:
                        is_method = True
                        break
                f = self.funcbody(is_method, full, line)
                return ('assign', [target], [f], line)
            if tx == 'local':
                self.p += 1
                if self.accept('function'):
                    name = self.expect_name()
                    return ('localfunc', name, self.funcbody(False, name, line), line)
                names = [self.expect_name()]
                while self.accept(','):
                    names.append(self.expect_name())
                exprs = self.exprlist() if self.accept('=') else []
                return ('local', names, exprs, line)
            self.err('unexpected keyword')
        e = self.suffixedexp()
        if self.check('=') or self.check(','):
            targets = [e]
            while self.accept(','):
                targets.append(self.suffixedexp())
            self.expect('=')
            exprs = self.exprlist()
            for t in targets:
                if t[0] not in ('name', 'index'):
                    self.err('cannot assign to this expression')
            return ('assign', targets, exprs, line)
        if e[0] not in ('call', 'method'):
            self.err('syntax error (statement is neither an assignment nor a call)')
        return ('callstat', e, line)

    def funcbody(self, is_method, name, line):
        self.expect('(')
        params, vararg = (['self'] if is_method else []), False
        if not self.check(')'):
            while True:
                if self.accept('...'):
                    vararg = True
                    break
                params.append(self.expect_name())
                if not self.accept(','):
                    break
        self.expect(')')
        body = self.block()
        self.expect('end')
        return ('func', params, vararg, body, name, line)

    # ---- expressions
    def exprlist(self):
        out = [self.expr()]
        while self.accept(','):
            out.append(self.expr())
        return out

    def primaryexp(self):
        kind, tx, line = self.peek()
        if kind == 'name':
            self.p += 1
            return ('name', tx, line)
        if self.accept('('):
            e = self.expr()
            self.expect(')')
            return ('paren', e)
        self.err('unexpected symbol')

    def suffixedexp(self):
        e = self.primaryexp()
        while True:
            kind, tx, line = self.peek()
            if self.check('.'):
                self.p += 1
                e = ('index', e, ('str', self.expect_name()), line)
            elif self.check('['):
                self.p += 1
                k = self.expr()
                self.expect(']')
                e = ('index', e, k, line)
            elif self.check(':'):
                self.p += 1
                name = self.expect_name()
                e = ('method', e, name, self.callargs(), line)
            elif self.check('(') or self.check('{') or kind == 'str':
                e = ('call', e, self.callargs(), line)
            else:
                return e

    def callargs(self):
        kind, tx, _ = self.peek()
        if kind == 'str':
            self.p += 1
            return [('str', tx)]
        if self.check('{'):
            return [self.table()]
        self.expect('(')
        args = [] if self.check(')') else self.exprlist()
        self.expect(')')
        return args

    def table(self):
        line = self.line()
        self.expect('{')
        items = []
        while not self.check('}'):
            kind, tx, _ = self.peek()
            if self.check('['):
                self.p += 1
                k = self.expr()
                self.expect(']')
                self.expect('=')
                items.append(('hash', k, self.expr()))
            elif kind == 'name' and self.peek(1)[0] == 'op' and self.peek(1)[1] == '=':
                self.p += 2
                items.append(('hash', ('str', tx), self.expr()))
            else:
                items.append(('pos', None, self.expr()))
            if not (self.accept(',') or self.accept(';')):
                break
        self.expect('}')
        return ('table', items, line)

    def simpleexp(self):
        kind, tx, line = self.peek()
        if kind == 'num':
            self.p += 1
            return ('num', tx)
        if kind == 'num64':
            self.p += 1
            return ('num64', tx)
        if kind == 'str':
            self.p += 1
            return ('str', tx)
        if kind == 'kw' and tx in ('nil', 'true', 'false'):
            self.p += 1
            return (tx,)
        if self.check('...'):
            self.p += 1
            return ('vararg', line)
        if self.check('{'):
            return self.table()
        if self.check('function'):
            self.p += 1
            return self.funcbody(False, 'anonymous', line)
        return self.suffixedexp()

    def expr(self, limit=0):
        kind, tx, line = self.peek()
        if (kind == 'kw' and tx == 'not') or (kind == 'op' and tx in ('-', '#')):
            self.p += 1
            operand = self.expr(UNARY_PRI)
            if tx == '-' and operand[0] == 'num':
                left = ('num', -operand[1])
            else:
                left = ('un', tx, operand, line)
        else:
            left = self.simpleexp()
        while True:
            kind, tx, line = self.peek()
            if kind not in ('op', 'kw') or tx not in BINPRI:
                break
            lp, rp = BINPRI[tx]
            if lp <= limit:
                break
            self.p += 1
            right = self.expr(rp)
            if tx in ('and', 'or'):
                left = (tx, left, right)
            else:
                left = ('bin', tx, left, right, line)
        return left


def parse(src, name='<lua>'):
    return Parser(src, name).chunk()
