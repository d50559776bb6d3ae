"""The Lua-side surface (lua/), static checks: the generated ffi.cdef declares exactly the header's entry points and structs, the
generator is reproducible, the plug-in files exist under the reference's names, model.lua only calls symbols the library exports,
every file parses and reads no undeclared name.  (The files are EXECUTED by tests/test_luavm_cpu.py and tests/test_lua_host_gpu.py.)"""
import os
import re
import subprocess
import sys
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_ENCODERS = ['lf-ques', 'lf-ques-im', 'lf-ques-hist', 'lf-ques-im-hist', 'lf-att-ques-im-hist', 'hre-ques-hist',
                'hre-ques-im-hist', 'hrea-ques-im-hist', 'mn-ques-hist', 'mn-ques-im-hist', 'mn-att-ques-im-hist']


def header_symbols():
    h = open(os.path.join(ROOT, 'include', 'visdial_hip.h')).read()
    return set(re.findall(r'\b(vd_[a-z0-9_]+)\s*\(', h)), set(re.findall(r'\}\s*(vd_[a-z0-9_]+)\s*;', h))


def test_cdef_declares_the_whole_header():
    lua = open(os.path.join(ROOT, 'lua', 'visdial_ffi.lua')).read()
    cdef = lua[lua.index('ffi.cdef[['):lua.index(']]')]
    funcs, structs = header_symbols()
    assert set(re.findall(r'\b(vd_[a-z0-9_]+)\s*\(', cdef)) == funcs and len(funcs) >= 70
    assert set(re.findall(r'\}\s*(vd_[a-z0-9_]+)\s*;', cdef)) == structs and 'vd_model_params' in structs
    assert '#' not in cdef and 'extern' not in cdef and '/*' not in cdef        # nothing LuaJIT's parser rejects
    assert cdef.count('{') == cdef.count('}')
    listed = set(re.findall(r"'(vd_[a-z0-9_]+)'", lua[lua.index('M.symbols'):]))
    assert listed == funcs


def test_generator_is_reproducible():
    before = open(os.path.join(ROOT, 'lua', 'visdial_ffi.lua')).read()
    subprocess.check_call([sys.executable, os.path.join(ROOT, 'scripts', 'gen_lua_ffi.py')], stdout=subprocess.DEVNULL)
    assert open(os.path.join(ROOT, 'lua', 'visdial_ffi.lua')).read() == before, "lua/visdial_ffi.lua is stale: re-run the generator"


def test_plugin_files_exist_under_the_reference_names():
    for e in REF_ENCODERS:
        src = open(os.path.join(ROOT, 'lua', 'encoders', e + '.lua')).read()
        assert 'function encoderNet.model(params)' in src and 'return encoderNet' in src and "native = '%s'" % e in src
    for d in ('disc', 'gen'):
        src = open(os.path.join(ROOT, 'lua', 'decoders', d + '.lua')).read()
        for fn in ('decoderNet.model(params, enc)', 'decoderNet.forwardConnect(enc, dec, encOut, seqLen)',
                   'decoderNet.backwardConnect(enc, dec)'):
            assert fn in src, (d, fn)


def test_model_lua_calls_only_exported_symbols():
    src = open(os.path.join(ROOT, 'lua', 'model.lua')).read()
    funcs, structs = header_symbols()
    used = set(re.findall(r"vd\.call\('(vd_[a-z0-9_]+)'", src)) | set(re.findall(r'\bC\.(vd_[a-z0-9_]+)', src))
    assert used and used <= funcs, used - funcs
    assert set(re.findall(r"ffi\.new\('(vd_[a-z_]+)", src)) <= structs | {'vd_model'}
    for method in ('__init', 'trainIteration', 'forwardBackward', 'retrieveBatch', 'retrieve', 'predict'):
        assert ('function Model:%s(' % method) in src
    # every struct field model.lua assigns exists in the header's struct
    h = open(os.path.join(ROOT, 'include', 'visdial_hip.h')).read()
    body = h[h.index('typedef struct vd_model_params'):h.index('} vd_batch;')]
    for field in set(re.findall(r'\b[pb]\.([A-Za-z_]+)\s*=', src)):
        assert re.search(r'\b%s\b' % field, body), field


def test_model_lua_keeps_the_reference_script_contract():
    """What the reference's UNCHANGED train.lua / evaluate.lua / generate.lua read and write on the model object
    (train.lua:79-81,89,100-102,113,120; evaluate.lua:91; generate.lua:83) must exist in lua/model.lua."""
    src = open(os.path.join(ROOT, 'lua', 'model.lua')).read()
    code = '\n'.join(l.split('--')[0] for l in src.splitlines())          # comments stripped
    # the loss EMA is the GLOBAL `runningLoss` (train.lua:89 sets it, train.lua:113 prints it; model.lua:81-92)
    assert re.search(r'(?m)^\s*(if .* then )?runningLoss\s*=', code), "trainIteration must assign the global runningLoss"
    assert 'self.runningLoss' not in code and not re.search(r'local\s+runningLoss', code)
    # gen: the EMA is fed curLoss / numTokens (model.lua:76-85)
    assert re.search(r"decoder\s*==\s*'gen'\s*then\s*cur\s*=\s*curLoss\s*/", code)
    assert ":gt(0):sum()" in code
    # model.wrapperW is a real tensor that can be read AND written; written back before any device call
    assert "key == 'wrapperW'" in code and 'function Model:commitW()' in code
    for method in ('trainIteration', 'forwardBackward', 'retrieveBatch', 'generateAnswers'):
        body = code[code.index('function Model:%s(' % method):]
        body = body[:body.index('\nend\n')]
        assert 'self:commitW()' in body, method
        first_call = re.search(r"vd\.call\(", body)
        assert body.index('self:commitW()') < first_call.start(), method
    # fields / methods the scripts use
    for needle in ('self.optims = {learningRate', 'function Model:retrieve(', 'function Model:predict(',
                   'function Model:evaluate(', 'function Model:generateAnswers(', 'self.optims.learningRate = '):
        assert needle in code, needle
    # the three scripts need no edit: INTEGRATION.md must not tell the user to edit them any more
    integ = open(os.path.join(ROOT, 'INTEGRATION.md')).read()
    assert 'setFlatParameters(savedModel.modelW)' not in integ


def _lua_tokens(src):
    """identifiers / keywords of Lua source with comments and string literals removed"""
    out, i, n = [], 0, len(src)
    while i < n:
        ch = src[i]
        if src.startswith('--[[', i):
            i = src.index(']]', i) + 2
        elif src.startswith('--', i):
            j = src.find('\n', i)
            i = n if j < 0 else j
        elif src.startswith('[[', i):
            i = src.index(']]', i) + 2
        elif ch in '"\'':
            j = i + 1
            while src[j] != ch:
                j += 2 if src[j] == '\\' else 1
            i = j + 1
        elif ch.isalpha() or ch == '_':
            j = i
            while j < n and (src[j].isalnum() or src[j] == '_'):
                j += 1
            out.append(src[i:j])
            i = j
        else:
            if ch in '(){}':
                out.append(ch)
            i += 1
    return out


def test_lua_files_are_block_balanced():
    """no Lua interpreter exists here: at least every hand-written Lua file must have balanced blocks and brackets
    (function / do / if / repeat ... end / until), the commonest slip in code that cannot be run"""
    files = ['model.lua', 'model_ops.lua', 'vdnn.lua', 'visdial_ffi.lua'] + ['encoders/%s.lua' % e for e in REF_ENCODERS] + ['decoders/disc.lua', 'decoders/gen.lua']
    for name in files:
        toks = _lua_tokens(open(os.path.join(ROOT, 'lua', name)).read())
        depth = 0
        for t in toks:
            if t in ('function', 'do', 'if', 'repeat'):
                depth += 1
            elif t in ('end', 'until'):
                depth -= 1
            assert depth >= 0, name
        assert depth == 0, (name, depth)
        assert toks.count('(') == toks.count(')') and toks.count('{') == toks.count('}'), name


def test_lua_files_parse_and_use_no_undeclared_names():
    """tests/lua_lint.py: a full Lua 5.1 parser + scope walk in Python (no interpreter exists here).  Every hand-written and
    generated Lua file must parse, read no global except `torch` (everything else is a local, a parameter or required into a
    local) and create no global except the reference's `runningLoss` (train.lua:89,113 reads it)."""
    import glob
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import lua_lint
    # the checker itself: accepts the constructs the files use, rejects the slips that matter
    good = ("local a, b = 1, 2 local function f(x, ...) local t = {1, x = 3, [4] = 5; ...} return #t .. 's', -x ^ 2 end "
            "for i = 1, 10 do if i % 2 == 0 then a = a + i elseif i > 5 then b = b * 2 else break end end "
            "for k, v in pairs({}) do f(k, v) end repeat local z = 1 until z == 1 local s = [[long]] --[[ c ]] "
            "local o = {} function o.m(self) return self end function o:n(y) return self, y end f(1)(2) f{1} f'q' return a")
    assert lua_lint.check(good) == []
    for bad in ('local x = = 1', 'if a then', 'for i = 1 do end', 'f(', 'local t = {1, 2', 'x = 1 +', 'return 1 local a', 'a.b:c',
                'local function() end', 'x = 1 2', 'local s = "abc', 'function f() return end end'):
        with pytest.raises(lua_lint.LuaSyntaxError):
            lua_lint.check(bad)
    assert [n for n, _ in lua_lint.check('local a = 1\nprint(a, bb)\nlocal function g() return cc end')] == ['bb', 'cc']
    files = sorted(glob.glob(os.path.join(ROOT, 'lua', '*.lua')) + glob.glob(os.path.join(ROOT, 'lua', '*', '*.lua')))
    assert len(files) == 18
    for f in files:
        parser = lua_lint.Parser(open(f).read(), os.path.relpath(f, ROOT), lua_lint.LUA_GLOBALS)
        parser.chunk()                                            # LuaSyntaxError with file:line on any syntax slip
        reads = set(n for n, _ in parser.undeclared if n not in parser.assigned_globals)
        # (dump_param_order.lua runs inside the REFERENCE checkout under Torch7: `Model` is the global class the reference's model.lua defines)
        assert reads <= ({'torch', 'Model'} if f.endswith('dump_param_order.lua') else {'torch'}), (f, sorted(reads))
        assert parser.assigned_globals <= {'runningLoss'}, (f, sorted(parser.assigned_globals))


def test_lua_ffi_calls_pass_as_many_arguments_as_the_header_declares():
    """every vd.call('vd_x', ...) / C.vd_x(...) in the hand-written Lua files passes exactly the number of arguments of the
    prototype in include/visdial_hip.h (LuaJIT's ffi raises on a mismatch only when the line runs)"""
    import glob
    h = re.sub(r'/\*.*?\*/', '', open(os.path.join(ROOT, 'include', 'visdial_hip.h')).read(), flags=re.S)
    protos = {}
    for m in re.finditer(r'\b(vd_[a-z0-9_]+)\s*\(([^;{}]*?)\)\s*;', h):
        args = m.group(2).strip()
        protos[m.group(1)] = 0 if args in ('', 'void') else args.count(',') + 1
    assert len(protos) >= 80
    strip = lambda t: '\n'.join(l.split('--')[0] for l in t.splitlines())
    sites = 0
    for f in sorted(glob.glob(os.path.join(ROOT, 'lua', '*.lua')) + glob.glob(os.path.join(ROOT, 'lua', '*', '*.lua'))):
        if f.endswith('visdial_ffi.lua'):
            continue
        src = strip(open(f).read())
        for m in re.finditer(r"vd\.call\('(vd_[a-z0-9_]+)'|\bC\.(vd_[a-z0-9_]+)\(", src):
            name, via_call = (m.group(1), True) if m.group(1) else (m.group(2), False)
            j, depth, commas, seen = m.end(), 1, 0, False
            while depth > 0:
                ch = src[j]
                if ch in '([{':
                    depth += 1
                elif ch in ')]}':
                    depth -= 1
                elif ch == ',' and depth == 1:
                    commas += 1
                if depth > 0 and not ch.isspace():
                    seen = True
                j += 1
            n = commas if via_call else (commas + 1 if seen else 0)       # vd.call's first argument is the name
            assert name in protos, (f, name)
            assert n == protos[name], (os.path.relpath(f, ROOT), name, n, protos[name])
            sites += 1
    assert sites >= 80

