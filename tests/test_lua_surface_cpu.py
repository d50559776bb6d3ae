"""The Lua-side surface (lua/): no Lua interpreter exists in the build container, so what CAN be pinned is pinned --
the generated ffi.cdef declares exactly the header's entry points and structs, the generator is reproducible, the
plug-in files exist under the reference's names, and model.lua only calls symbols the library exports."""
import os
import re
import subprocess
import sys

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
