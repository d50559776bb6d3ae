"""tests/luavm -- the Lua 5.1 evaluator, its LuaJIT-style ffi and its Torch7 stub -- checked on their own (language semantics
against the Lua 5.1 manual's behaviour, ffi conversion rules against a gcc-built C library, tensor semantics against numpy), and the
WHOLE Lua host executed on the CPU against a dry library: lua/model.lua (every method), lua/model_ops.lua + vdnn.lua with all
11 x 2 plug-in pairs.  The dry library has no arithmetic (numbers are checked on the GPU, tests/test_lua_host_gpu.py) but it
bounds-checks every memset / memcpy, refuses host / freed / mistyped pointers, and validates uploaded batches -- so nil-index,
arity, stride, lifetime and 0/1-based mistakes in the Lua files fail here, without a GPU."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT, small_params
from lua_host import LuaHost, first
from luavm import LuaError, new_vm, to_lua, to_py
from luavm.parse import LuaSyntaxError, parse
from visdial_amd.dataloader import SyntheticDataloader
from visdial_amd.opts import derive

ALL_ENC = ['lf-ques', 'lf-ques-im', 'lf-ques-hist', 'lf-ques-im-hist', 'lf-att-ques-im-hist', 'hre-ques-hist',
           'hre-ques-im-hist', 'hrea-ques-im-hist', 'mn-ques-hist', 'mn-ques-im-hist', 'mn-att-ques-im-hist']


def run(src, *args):
    return [to_py(v) for v in new_vm().dostring(src, *args)]


# ------------------------------------------------------------------------------------------------------------------- language
def test_closures_capture_a_fresh_variable_per_iteration():
    assert run('''local f = {}
                  for i = 1, 3 do local j = i * 10; f[i] = function() j = j + 1; return i, j end end
                  local a, b = f[1](); local c, d = f[1](); local e, g = f[3]()
                  return a, b, c, d, e, g''') == [1, 11, 1, 12, 3, 31]
    assert run('''local fns = {}
                  local i = 1
                  while i <= 2 do local k = i; fns[i] = function() return k end; i = i + 1 end
                  return fns[1](), fns[2]()''') == [1, 2]
    assert run('''local function counter() local n = 0; return function() n = n + 1; return n end end
                  local a, b = counter(), counter(); a(); a()
                  return a(), b()''') == [3, 1]


def test_multiple_values_varargs_and_truncation():
    assert run('''local function f() return 1, 2, 3 end
                  local t = {f(), f()}            -- first call truncated, last expanded
                  local a, b, c, d = f()
                  local x = (f())
                  local function g(...) return select('#', ...), ... end
                  return #t, a, b, c, d, x, (g(7, 8)), g(nil, nil)''') == [4, 1, 2, 3, None, 1, 2, 2, None, None]
    assert run('''local function v(...) local a, b = ...; local t = {...}; return a, b, #t, select(2, ...) end
                  return v(5, 6, 7)''') == [5, 6, 3, 6, 7]
    assert run('return unpack({1, 2, 3}, 2)') == [2, 3]


def test_metatables_and_operators():
    r = run('''local V = {}
               V.__index = V
               V.__add = function(a, b) return setmetatable({x = a.x + b.x}, V) end
               V.__eq = function(a, b) return a.x == b.x end
               V.__lt = function(a, b) return a.x < b.x end
               V.__le = function(a, b) return a.x <= b.x end
               V.__call = function(self, k) return self.x * k end
               V.__tostring = function(self) return 'V(' .. self.x .. ')' end
               V.__concat = function(a, b) return 'cat' end
               V.__unm = function(a) return setmetatable({x = -a.x}, V) end
               V.__newindex = function(t, k, v) rawset(t, k, v * 2) end
               function V:double() return self.x * 2 end
               local a, b = setmetatable({x = 1}, V), setmetatable({x = 2}, V)
               a.y = 5
               return (a + b).x, a == b, a ~= b, a < b, a <= b, a > b, b(10), tostring(a), a .. b, (-a).x, a:double(), a.y,
                      rawget(a, 'zzz'), getmetatable(a) == V''')
    assert r == [3, False, True, True, True, False, 20, 'V(1)', 'cat', -1, 2, 10, None, True]
    assert run('''local t = setmetatable({}, {__index = function(t, k) return k .. '?' end})
                  local u = setmetatable({}, {__index = t})
                  return u.abc, #u''') == ['abc?', 0]


def test_arithmetic_strings_and_number_formatting():
    assert run("return 7 / 2, 7 % 3, -7 % 3, 7 % -3, 2 ^ 10, 2 ^ 0.5, 10 / 5, '10' + 5, 1e3, 3 == 3.0, 0.1 + 0.2 == 0.3") == \
        [3.5, 1, 2, -2, 1024, 2 ** 0.5, 2.0, 15, 1000.0, True, False]
    assert run("return tostring(10 / 5), tostring(1e15), tostring(0.1), tostring(-0.0 + 3), tostring(1 / 0), tostring(2 ^ 53), 1 .. ''") == \
        ['2', '1e+15', '0.1', '3', 'inf', '9.007199254741e+15', '1']
    assert run("return #'abc', 'a' .. 1 .. 2.5, 'abc' < 'abd', 'Z' < 'a', ('x'):rep(3), ('hello'):sub(2, -2), ('hello'):sub(-3)") == \
        [3, 'a12.5', True, True, 'xxx', 'ell', 'llo']
    assert run("return tonumber('0x10'), tonumber('  12  '), tonumber('1e2'), tonumber('abc'), tonumber('10', 2), tonumber(nil)") == \
        [16, 12, 100.0, None, 2, None]
    assert run("return math.floor(-3.5), math.max(1, 5, 3), math.huge > 1e308, math.floor(7 / 2) * 4, 10 == '10'") == [-4, 5, True, 12, False]


def test_string_library_and_patterns():
    assert run("return string.format('%5.2f|%-4d|%s|%q|%x|%5s|%%|%g', 3.14159, 7, nil, 'a\"b', 255, 'ab', 0.5)") == \
        [' 3.14|7   |nil|"a\\"b"|ff|   ab|%|0.5']
    assert run("return string.match('mn-att-ques-im-hist', 'mn'), string.match('lf-ques', '^hre'), string.match('img_embed.W', '^img_embed%.')") == \
        ['mn', None, 'img_embed.']
    assert run("return string.find('hello world', 'o w'), string.find('a.b', '.', 1, true), string.find('abc', '(b)(c)')") == \
        [5, 2, 2, 3, 'b', 'c']                      # every call but the last is truncated to one value
    assert run("return ('COCO_train2014_000000123456'):match('000%d+'), ('k1=v1, k2=v2'):gsub('(%w+)=(%w+)', '%2=%1')") == \
        ['000000123456', 'v1=k1, v2=k2', 2]
    assert run("local t = {} for w in ('one two  three'):gmatch('%a+') do t[#t + 1] = w end return t") == [['one', 'two', 'three']]
    assert run("return ('x'):byte(), string.char(72, 105), ('AbC'):lower(), ('%d'):format(3.0)") == [120, 'Hi', 'abc', '3']


def test_tables_length_insert_remove_sort_next():
    assert run('''local t = {10, 20, 30}
                  table.insert(t, 40); table.insert(t, 1, 5)
                  local r = table.remove(t, 2)
                  t[#t + 1] = 99
                  t[#t] = nil
                  return #t, r, table.concat(t, ','), table.remove(t), #t''') == [4, 10, '5,20,30,40', 40, 3]
    assert run('''local t = {}
                  t[3] = 'c'; t[2] = 'b'; t[1] = 'a'        -- filled backwards: all three become the array part
                  local n = 0 for _ in pairs(t) do n = n + 1 end
                  return #t, n, t[1.0], t['1']''') == [3, 3, 'a', None]
    assert run('''local t = {{id = 1, s = 2}, {id = 2, s = 9}, {id = 3, s = 2}}
                  table.sort(t, function(a, b) if a.s ~= b.s then return a.s > b.s end return a.id < b.id end)
                  return t[1].id, t[2].id, t[3].id''') == [2, 1, 3]
    assert run('''local t = {a = 1, b = 2, 10}
                  local keys = {}
                  for k, v in pairs(t) do keys[#keys + 1] = tostring(k) end
                  table.sort(keys)
                  local k1 = next({})
                  t[true] = 'yes'; t[1] = 'one'
                  return table.concat(keys, ','), k1, t[true], t[1]''') == ['1,a,b', None, 'yes', 'one']
    assert run("local t = {n = 0}; for i, v in ipairs({5, 6, nil, 8}) do t.n = t.n + v end return t.n") == [11]


def test_errors_pcall_and_positions():
    r = run('''local ok1, e1 = pcall(error, {code = 7})
               local ok2, e2 = pcall(function() local t = nil; return t.x end)
               local ok3, e3 = pcall(function() return 1 + {} end)
               local ok4, e4 = pcall(function() undefined_function() end)
               local ok5, e5 = pcall(function() error('msg', 0) end)
               local ok6, e6 = pcall(function() error('located') end)
               local ok7, e7 = pcall(function() for i = 1, 'x' do end end)
               local ok8 = pcall(function() return #nil end)
               return ok1, e1.code, ok2, e2, ok3, e3, ok4, e4, e5, e6, ok7, ok8, select(2, pcall(assert, false, 'why')), (select(2, pcall(assert, nil)))''')
    assert r[0] is False and r[1] == 7
    assert "attempt to index a nil value (variable 't')" in r[3]
    assert 'attempt to perform arithmetic on a table value' in r[5]
    assert "attempt to call a nil value (variable 'undefined_function')" in r[7]
    assert r[8] == 'msg' and r[9].endswith(':6: located')
    assert r[10] is False and r[11] is False and r[12] == 'why' and r[13] == 'assertion failed!'
    with pytest.raises(LuaSyntaxError):
        parse('local x = = 1')
    with pytest.raises(LuaSyntaxError):
        parse('for i = 1 do end')
    with pytest.raises(LuaSyntaxError):
        parse('return 1 x = 2')


def test_control_flow_scoping_and_goto_free_idioms():
    assert run('''local out = {}
                  for i = 10, 1, -3 do out[#out + 1] = i end
                  for i = 1, 0 do out[#out + 1] = 'never' end
                  local i = 0
                  repeat local done = i >= 2; i = i + 1 until done
                  while true do i = i + 10; if i > 20 then break end end
                  local x = 1
                  do local x = 2; out[#out + 1] = x end
                  out[#out + 1] = x
                  local function fact(n) if n <= 1 then return 1 else return n * fact(n - 1) end end
                  return table.concat(out, ','), i, fact(10), (nil or false) == false, (false and 1) == false, nil and 1, 0 and 'zero is true\'''') == \
        ['10,7,4,1,2,1', 23, 3628800, True, True, None, 'zero is true']


def test_dofile_require_and_globals(tmp_path):
    (tmp_path / 'mod.lua').write_text("local M = {}; function M.twice(x) return 2 * x end; counter = (counter or 0) + 1; return M\n")
    (tmp_path / 'sub').mkdir()
    (tmp_path / 'sub' / 'init.lua').write_text("return {name = ...}\n")
    vm = new_vm(search=[str(tmp_path)])
    r = vm.dostring('''local a = require 'mod'; local b = require('mod'); local c = dofile('mod.lua')
                       return a == b, a == c, counter, a.twice(4), require('sub').name, package.loaded.mod == a,
                              select(2, pcall(require, 'nope')), type(_G), _G.counter''')
    assert [to_py(x) for x in r][:6] == [True, False, 2, 8, 'sub', True]
    assert "module 'nope' not found" in r[6] and r[7] == 'table' and r[8] == 2


# ------------------------------------------------------------------------------------------------------------------------ ffi
CSRC = r'''
#include <stdint.h>
#include <string.h>
typedef struct { int32_t a; float b; const int32_t* p; int64_t big; char tag[8]; } rec_t;
int add3(int a, float b, double c) { return (int)(a + b + c); }
int64_t big(int64_t x) { return x * 2; }
float sum_floats(const float* x, int n) { float s = 0; for (int i = 0; i < n; ++i) s += x[i]; return s; }
void fill_ints(int32_t* x, int n, int v) { for (int i = 0; i < n; ++i) x[i] = v + i; }
int rec_sum(const rec_t* r) { return r->a + (int)r->b + (r->p ? r->p[1] : 0) + (int)r->big + r->tag[0]; }
const char* hello(void) { return "hello"; }
int is_null(const void* p) { return p == 0; }
int first_byte(const void* p) { return *(const unsigned char*)p; }
void set_ptr(void** out, void* v) { *out = v; }
'''
CDEF = '''typedef struct { int32_t a; float b; const int32_t* p; int64_t big; char tag[8]; } rec_t;
int add3(int a, float b, double c); int64_t big(int64_t x); float sum_floats(const float* x, int n);
void fill_ints(int32_t* x, int n, int v); int rec_sum(const rec_t* r); const char* hello(void); int is_null(const void* p);
int first_byte(const void* p); void set_ptr(void** out, void* v); static const int MAGIC = 42;'''


@pytest.fixture(scope='module')
def clib(tmp_path_factory):
    d = tmp_path_factory.mktemp('clib')
    (d / 't.c').write_text(CSRC)
    so = str(d / 'libt.so')
    subprocess.check_call(['gcc', '-shared', '-fPIC', '-O1', '-o', so, str(d / 't.c')])
    return so


def ffi_run(clib, body):
    vm = new_vm()
    vm.globals.set('LIB', clib)
    vm.globals.set('CDEF', CDEF)
    return [to_py(v) for v in vm.dostring("local ffi = require 'ffi'; ffi.cdef(CDEF); local L = ffi.load(LIB)\n" + body)]


def test_ffi_calls_structs_arrays_and_pointer_arithmetic(clib):
    r = ffi_run(clib, '''
        local x = ffi.new('float[4]', {1.5, 2.5, 3, 4})
        local ints = ffi.new('int32_t[?]', 5)
        L.fill_ints(ints, 5, 100)
        local r = ffi.new('rec_t')
        r.a = 3; r.b = 2.9; r.p = ints; r.big = 10; r.tag = 'A'
        local p = ffi.cast('int32_t*', ints) + 2
        local vp = ffi.new('void*[1]')
        L.set_ptr(vp, ints)
        local back = ffi.cast('int32_t*', vp[0])
        return L.add3(1, 2.5, 3.25), tonumber(L.big(2 ^ 40)), type(L.big(1)), L.sum_floats(x, 4), L.sum_floats(x + 1, 2), ints[0], ints[4], p[0], p[-1],
               L.rec_sum(r), ffi.string(L.hello()), L.MAGIC, r.a, r.b, tonumber(r.big), ffi.string(r.tag), back[3], (p - ints), ffi.sizeof('rec_t'),
               back == ints, ffi.cast('void*', nil) == nil, L.is_null(nil), L.first_byte('Zz'), tostring(L.big(3)), ffi.sizeof(ints)''')
    assert r == [6, 2 ** 41, 'cdata', 11.0, 5.5, 100, 104, 102, 101, 3 + 2 + 101 + 10 + 65, 'hello', 42, 3, pytest.approx(2.9, rel=1e-6), 10, 'A', 103, 2,
                 32, True, True, 1, 90, '6LL', 20]


def test_ffi_conversion_rules_are_luajit_s(clib):
    """doc/ext_ffi_semantics: pointer compatibility (no cast): same size + class of the pointee, qualifiers may be added not dropped,
    void* converts both ways; numbers never become pointers; 64-bit integers are boxed"""
    bad = ffi_run(clib, '''
        local ints, floats = ffi.new('int32_t[4]'), ffi.new('float[4]')
        local cf = ffi.cast('const float*', floats)
        local errs = {}
        local function try(f) local ok, e = pcall(f); errs[#errs + 1] = ok and 'ok' or e end
        try(function() L.sum_floats(ints, 4) end)                       -- int32_t* -> const float*
        try(function() L.fill_ints(ffi.cast('const int32_t*', ints), 4, 0) end)   -- drops const
        try(function() L.sum_floats(cf, 4) end)                         -- const -> const: fine
        try(function() L.sum_floats(ffi.cast('void*', floats), 4) end)  -- void* -> T*: fine
        try(function() L.sum_floats(12345, 4) end)                      -- number -> pointer
        try(function() L.sum_floats(floats) end)                        -- arity
        try(function() L.fill_ints(ffi.cast('uint32_t*', ints), 4, 0) end)      -- signedness is ignored: fine
        try(function() return ints[4] end)                              -- luavm is stricter: bounds
        try(function() local r = ffi.new('rec_t'); r.nope = 1 end)
        try(function() L.add3(true, 1, 1) end)
        try(function() for i = 0, L.big(2) do end end)                  -- a boxed int64 is not a Lua number
        try(function() L.first_byte(ffi.cast('const void*', 'str')) end)      -- string -> const void* cast: fine
        try(function() L.set_ptr(ffi.new('void*[1]'), 'str') end)       -- string -> non-const void*
        return errs''')[0]
    assert 'cannot convert' in bad[0] and "'const float *'" in bad[0]
    assert 'cannot convert' in bad[1]
    assert bad[2] == 'ok' and bad[3] == 'ok'
    assert 'cannot convert' in bad[4]
    assert 'wrong number of arguments' in bad[5]
    assert bad[6] == 'ok'
    assert 'outside' in bad[7]
    assert "no member named 'nope'" in bad[8]
    assert "cannot convert 'boolean'" in bad[9]
    assert "'for' limit value must be a number" in bad[10]
    assert bad[11] == 'ok'
    assert 'cannot convert' in bad[12]


def test_ffi_gc_finalizers_run_once_and_at_close(clib):
    vm = new_vm()
    log = []
    vm.globals.set('note', lambda p: log.append(type(p).__name__))
    vm.dostring('''local ffi = require 'ffi'
                   local a = ffi.gc(ffi.cast('void*', 16), note)        -- collected when the chunk returns
                   keep = ffi.gc(ffi.cast('void*', 32), note)           -- a global: alive until close
                   local c = ffi.gc(ffi.cast('void*', 48), note); ffi.gc(c, nil)   -- finaliser removed''')
    assert log == ['CPointer']
    vm.close()
    assert log == ['CPointer', 'CPointer']


def test_ffi_struct_layout_matches_gcc_for_the_real_header(tmp_path):
    """sizeof / offsetof of vd_batch, vd_model_params, vd_lstm2_*_t as luavm lays them out == what gcc does with include/visdial_hip.h"""
    src = tmp_path / 'lay.c'
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "visdial_hip.h"\nint main(void) {\n'
                   'printf("%zu %zu %zu %zu %zu %zu %zu %zu\\n", sizeof(vd_batch), sizeof(vd_model_params), sizeof(vd_lstm2_fwd_t), sizeof(vd_lstm2_bwd_t),'
                   ' offsetof(vd_batch, option_out), offsetof(vd_model_params, dropout), offsetof(vd_model_params, seed), offsetof(vd_lstm2_fwd_t, nact));\nreturn 0; }\n')
    exe = str(tmp_path / 'lay')
    subprocess.check_call(['gcc', '-I', os.path.join(ROOT, 'include'), '-o', exe, str(src)])
    want = [int(x) for x in subprocess.check_output([exe]).split()]
    vm = new_vm(search=[os.path.join(ROOT, 'lua')])
    os.environ['VD_LIB_PATH'] = os.path.join(ROOT, 'visdial_amd', 'libvisdial_hip.so')
    vm.dostring("dofile('visdial_ffi.lua')")
    f = vm.ffi
    off = lambda st, name: f.typedefs[st].index[name][1]
    got = [f.typedefs['vd_batch'].size, f.typedefs['vd_model_params'].size, f.typedefs['vd_lstm2_fwd_t'].size, f.typedefs['vd_lstm2_bwd_t'].size,
           off('vd_batch', 'option_out'), off('vd_model_params', 'dropout'), off('vd_model_params', 'seed'), off('vd_lstm2_fwd_t', 'nact')]
    assert got == want


# ---------------------------------------------------------------------------------------------------------------------- torch
def test_torch_stub_has_torch7_semantics():
    vm = new_vm()
    r = [to_py(v) for v in vm.dostring('''
        local t = torch.IntTensor(2, 3)
        for i = 1, 2 do for j = 1, 3 do t[i][j] = 10 * i + j end end
        local tt = t:t()
        local okv, ev = pcall(function() return tt:view(-1) end)                    -- view of a transposed tensor
        local c = tt:contiguous()
        local same = t:int() == t and t:contiguous() == t                         -- no copy when nothing changes
        local l = t:long()
        l[1][1] = 99                                                              -- a conversion is a copy
        local n = t:narrow(2, 2, 2)
        n:fill(0)                                                                 -- a narrow is a view
        local m = torch.ones(2, 2):byte()
        m[1][2] = 0
        local f = torch.FloatTensor({1.5, -2, 3})
        local idx = torch.LongTensor({3, 1})
        local s, si = torch.Tensor({3, 1, 2}):sort(1, true)
        return c:view(-1), okv, same, t[1][1], t, m:sum(), f:gt(0):sum(), torch.type(f:gt(0)), f:sum(), f:mean(), t:size(2), t:dim(), t:nElement(),
               #t:size(), f:index(1, idx), torch.repeatTensor(m, 2, 1):size(1), torch.totable(m), s, si, f[{{2, 3}}], torch.type(torch.Tensor(1)),
               t:stride(1), tt:stride(1), tt:isContiguous(), torch.LongTensor{5}[1]''')]
    assert r[0].tolist() == [11, 21, 12, 22, 13, 23] and r[1] is False and r[2] is True and r[3] == 11
    assert r[4].tolist() == [[11, 0, 0], [21, 0, 0]]
    assert r[5:14] == [3, 2, 'torch.ByteTensor', 2.5, pytest.approx(2.5 / 3), 3, 2, 6, 2]
    assert r[14].tolist() == [3.0, 1.5] and r[15] == 4 and r[16] == [[1, 0], [1, 1]]
    assert r[17].tolist() == [3, 2, 1] and r[18].tolist() == [1, 3, 2] and r[19].tolist() == [-2, 3]
    assert r[20:] == ['torch.DoubleTensor', 3, 1, False, 5]


def test_torch_stub_strictness_poison_and_data_pointers():
    vm = new_vm()
    r = vm.dostring('''local ffi = require 'ffi'
                       local t = torch.FloatTensor(4)              -- uninitialised memory is poison here, not zeros
                       local nan = t[1] ~= t[1]
                       local i = torch.IntTensor({1, 2, 3, 4})
                       local p = i:narrow(1, 2, 2):data()          -- typed int* at the storage offset
                       local v = p[0]
                       local dangling = torch.IntTensor({7, 8, 9}):data()     -- the tensor is garbage as soon as the statement ends
                       return nan, v, tostring(ffi.typeof(p)), dangling[0], select(2, pcall(function() return i[5] end)),
                              select(2, pcall(function() return i:cuda() end))''')
    assert r[0] is True and r[1] == 2 and 'int' in r[2]
    assert r[3] == 0x7F7F7F7F                                     # use-after-free reads the poison, deterministically
    assert 'out of range' in r[4] and 'no cutorch' in r[5]


def test_torch_class_builds_constructors_and_inheritance():
    r = run('''local baseMt
               do                                                   -- the locals are the METATABLES; the constructors are globals
                   local Base = torch.class('Base')
                   function Base:__init(x) self.x = x end
                   function Base:get() return self.x end
                   local Derived, parent = torch.class('pkg.Derived', 'Base')
                   function Derived:__init(x) parent.__init(self, x * 2) end
                   function Derived:more() return self:get() + 1 end
                   baseMt = Base
                   same = parent == Base
               end
               local b, d = Base(5), pkg.Derived(5)
               return b:get(), d:get(), d:more(), torch.type(d), torch.typename(b), torch.type({}), torch.isTypeOf(d, 'Base'), same''')
    assert r == [5, 10, 11, 'pkg.Derived', 'Base', 'table', True, True]


# ------------------------------------------------------------------------------- the Lua host of this repo, on the dry library
@pytest.mark.parametrize("enc", ALL_ENC)
@pytest.mark.parametrize("dec", ['disc', 'gen'])
def test_every_lua_composed_plugin_pair_executes_and_frees_its_activations(enc, dec):
    p = derive(small_params(encoder=enc, decoder=dec))
    host = LuaHost(p, dry=True)
    m = host.model_ops()
    dl = SyntheticDataloader(p, seed=3)
    batch = dl.getTrainBatch(p)
    lua_batch = host.batch(batch)
    host.invoke(m, 'forwardBackward', lua_batch)
    for k, v in lua_batch.items():
        np.testing.assert_array_equal(to_py(v), batch[k], err_msg='%s was modified by the step' % k)
    live1 = host.dry.bytes_allocated
    host.invoke(m, 'evaluate')
    host.invoke(m, 'forwardBackward', host.batch(batch), True)
    host.invoke(m, 'training')
    for _ in range(3):
        host.invoke(m, 'trainIteration', host.dataloader(dl))
    assert host.dry.bytes_allocated <= 1.2 * live1                 # step-scoped activations: no growth over iterations
    names = [c[0] for c in host.dry.calls]
    assert names.count('vd_clamp_adam') == 3 and 'vd_lstm_forward' in names and 'vd_lstm_backward' in names
    assert ('vd_score_ce' in names) == (dec == 'disc') and ('vd_logsoftmax_nll' in names) == (dec == 'gen')
    if dec == 'disc':
        host.invoke(m, 'evaluate')
        host.get(m, 'params').set('useGt', True)
        r = to_py(first(host.invoke(m, 'retrieveBatch', host.batch(batch))))
        assert r.shape == (batch['ques_fwd'].shape[0], p['maxQuesCount']) and r.dtype == np.float64
    # an all-IntTensor batch (answer_ind too) must survive the step unchanged: `:int()` of an IntTensor is the same tensor
    ints = dict.fromkeys(('ques_fwd', 'hist', 'options', 'answer_in', 'answer_out', 'answer_ind'), 'Int')
    ints['img_feat'] = 'Float'
    b2 = host.batch(batch, ints)
    host.invoke(m, 'forwardBackward', b2)
    for k, v in b2.items():
        np.testing.assert_array_equal(to_py(v), batch[k], err_msg=k)
    host.close()


@pytest.mark.parametrize("enc,dec", [('mn-att-ques-im-hist', 'disc'), ('lf-ques', 'gen'), ('hre-ques-im-hist', 'disc'), ('lf-ques-im-hist', 'gen')])
def test_the_model_level_lua_host_executes_every_method(enc, dec):
    p = derive(small_params(encoder=enc, decoder=dec))
    host = LuaHost(p, dry=True)
    m = host.model()
    dl = SyntheticDataloader(p, seed=5, num_threads=5)
    D = host.dataloader(dl)
    lr0 = host.get(m, 'optims', 'learningRate')
    for it in range(3):
        host.invoke(m, 'trainIteration', D)
    dry = host.dry
    mod = list(dry.models.values())[0]
    assert mod['steps'] == 3 and mod['uploaded'] == 4              # software pipeline: the first batch + one prefetch per step
    assert len(D.served) == 4                                      # ... and the dataloader was advanced exactly that often
    assert host.get(m, 'optims', 'learningRate') == pytest.approx(lr0 * p['lrDecayRate'] ** 3, rel=1e-6)
    run_loss = host.vm.globals.get('runningLoss')                  # the reference's GLOBAL (train.lua:89,113)
    cur = [4.0, 2.0, 4.0 / 3]
    if dec == 'gen':
        cur = [c / max(int((b['answer_out'] > 0).sum()), 1) for c, b in zip(cur, D.served)]
    want = cur[0]
    for c in cur[1:]:
        want = 0.95 * want + 0.05 * c
    assert run_loss == pytest.approx(want, rel=1e-6)
    # an evaluation between two training steps replaces the prefetched batch: the next step re-uploads, the dataloader is not advanced twice
    host.invoke(m, 'forwardBackward', host.batch(dl.getTrainBatch(p)), True)
    served = len(D.served)
    host.invoke(m, 'trainIteration', D)
    assert len(D.served) == served + 1 and mod['uploaded'] == 4 + 1 + 2
    # wrapperW: a host FloatTensor checked out of the library and committed before the next device call
    w = host.get(m, 'wrapperW')
    n = sum(int(np.prod(s)) for _, s, _ in mod['spec'])
    assert w.tn == 'Float' and w.a.shape == (n,)
    host.invoke(w, 'fill', 0.25)
    host.invoke(m, 'forwardBackward', host.batch(dl.getTrainBatch(p)), True)
    assert all(float(v.min()) == 0.25 == float(v.max()) for v in mod['W'].values())
    # hre-*: the flat vector is in the REFERENCE's getParameters() order (image Linear right behind the embedding)
    order = [t['name'] for t in to_py(first(host.invoke(m, 'tensors')))]
    if enc.startswith('hre'):
        assert order[:3] == ['embed', 'img_embed.W', 'img_embed.b']
    assert sorted(order) == sorted(n_ for n_, _, _ in mod['spec'])
    host.invoke(m, 'setDropoutMasks', to_lua(host.vm, {'fuse': np.ones(8, np.uint8)}))
    if dec == 'disc':
        D.fields['unique_img_val'] = to_lua(host.vm, list(range(101, 106)))
        D.fields['val_num_rounds'] = to_lua(host.vm, [p['maxQuesCount']] * 5)
        k = len(D.served)
        recs = to_py(first(host.invoke(m, 'retrieve', D, 'val')))
        gt = np.concatenate([np.asarray(b['answer_ind']).reshape(-1) for b in D.served[k:]])     # the dry library ranks option k at place k
        assert len(recs) == 5 * p['maxQuesCount'] and recs[0]['image_id'] == 101 and recs[0]['round_id'] == 1
        assert [r['ranks'] for r in recs] == [float(x) for x in gt]
        pred = to_py(first(host.invoke(m, 'predict', D, 'val')))
        assert pred[0]['ranks'] == [float(i) for i in range(1, p['numOptions'] + 1)]
    first(host.invoke(m, 'evaluate', host.dataloader(SyntheticDataloader(p, seed=5, num_threads=5)), 'val'))
    avail = host.vm.call(host.get(m, 'commAvailable'))
    assert avail[0] is True and avail[1] == 22203
    assert to_py(first(host.vm.call(host.get(m, 'commStats'))))['calls'] == 0
    token = first(host.vm.call(host.get(m, 'commUniqueId')))
    host.invoke(m, 'initComm', 1, 2, token)
    host.invoke(m, 'trainIteration', D)
    assert to_py(first(host.vm.call(host.get(m, 'commStats')))) == {'bucket1': 100, 'bucket2': 50, 'overlapped': True, 'calls': 1}
    assert [c[0] for c in dry.calls[-8:]].count('vd_model_allreduce_grads') == 1 and dry.comm == (1, 2)
    upd = [c for c in dry.calls if c[0] == 'vd_model_update'][-1]
    assert upd[1][1] == 0.5                                         # the 1 / world average goes into clamp + adam
    with pytest.raises(LuaError, match='128 bytes'):
        host.invoke(m, 'initComm', 0, 2, 'short')
    del m, mod
    host.close()
    assert not dry.models                                           # ffi.gc(handle, vd_model_destroy) ran


@pytest.mark.parametrize("enc", ALL_ENC)
def test_lua_flat_vector_is_in_the_executed_reference_s_order(enc):
    """lua/model.lua:Model:tensors() -- the order behind model.wrapperW / torch.save(modelW) -- equals the order
    wrapper:getParameters() produced when the reference's files were executed (tests/golden/reference_param_order.json) and
    visdial_amd/t7.py:reference_order, so a .t7 written by either host loads in the other (and in Torch7, as far as derived)."""
    import json
    from oracle import visdial_oracle as vo
    from visdial_amd import t7
    d = json.load(open(os.path.join(ROOT, 'tests', 'golden', 'reference_param_order.json')))
    cases = [('disc', {})] + [(v['decoder'], v['params']) for v in d['variants'] if v['encoder'] == enc]
    for dec, extra in cases:
        p = derive(small_params(encoder=enc, decoder=dec, **extra))
        host = LuaHost(p, dry=True)
        m = host.model()
        names = [t['name'] for t in to_py(first(host.invoke(m, 'tensors')))]
        want = d['encoder'][enc] + d['decoder'][dec] if not extra else [v['order'] for v in d['variants']
                                                                         if v['encoder'] == enc and v['params'] == extra][0]
        assert names == want
        assert names == [n for n, _, _ in t7.reference_order(enc, vo.param_spec(enc, dec, p))]
        flat = first(host.invoke(m, 'getFlatParameters'))                  # ... and the flat vector has every tensor exactly once
        assert flat.a.size == sum(int(np.prod(s)) for _, s, _ in vo.param_spec(enc, dec, p))


def test_lua_generate_answers_runs_beam_search_and_sampling():
    from test_dataloader_cpu import raw_dataset
    from visdial_amd.dataloader import Dataloader
    from visdial_amd.opts import default_params
    info, raw, img = raw_dataset(np.random.RandomState(2), n=3, R=3, MQ=5, MA=4, V=20, O=4, nopt=12, F=8)
    raw = {k.replace('_train', '_val'): v for k, v in raw.items()}
    img = {k.replace('_train', '_val'): v for k, v in img.items()}
    info['unique_img_val'] = info.pop('unique_img_train')
    p = derive(default_params(encoder='lf-ques-im-hist', decoder='gen', embedSize=12, rnnHiddenSize=32, imgFeatureSize=8, imgEmbedSize=8,
                              numLayers=2, batchSize=1, learningRate=1e-3, gpuid=0))
    dl = Dataloader(seed=1).from_arrays(info, raw, img, p, ['val'])
    for k in ('vocabSize', 'maxQuesCount', 'maxQuesLen', 'maxAnsLen'):
        p[k] = getattr(dl, k)
    host = LuaHost(p, dry=True)
    m = host.model()
    D = host.dataloader(dl)
    D.fields['word2ind'] = to_lua(host.vm, dict(dl.word2ind))
    D.fields['ind2word'] = to_lua(host.vm, {int(k): v for k, v in dl.ind2word.items()})
    D.fields['numThreads'] = to_lua(host.vm, {'val': 3})
    out = to_py(first(host.invoke(m, 'generateAnswers', D, 'val', to_lua(host.vm, dict(beamSize=3, beamLen=6, maxThreads=2)))))
    assert len(out) == 2 and len(out[0]['dialog']) == 3 and out[0]['dialog'][0]['answer'].startswith('<START>')
    smp = to_py(first(host.invoke(m, 'generateAnswers', D, 'val', to_lua(host.vm, dict(sampleWords=1, temperature=0.7, beamLen=5, maxThreads=1)))))
    assert len(smp[0]['dialog'][0]['answer'].split()) == 6
    host.close()


def test_the_dry_library_catches_the_mistakes_it_is_there_for():
    p = derive(small_params(encoder='lf-ques', decoder='disc'))
    host = LuaHost(p, dry=True)
    host.vm.dofile('model_ops.lua')
    errs = [to_py(v) for v in host.vm.dostring('''
        local ffi = require 'ffi'
        local vdnn = dofile('vdnn.lua')
        local vd = vdnn.vd
        local out = {}
        local function try(f) local ok, e = pcall(f); out[#out + 1] = ok and 'ok' or e end
        local x = vdnn.devFloats(8)
        try(function() vd.call('vd_memset', x, 0, 8 * 4 + 1, nil) end)                       -- one byte too many
        try(function() vd.call('vd_axpby', x, x, torch.FloatTensor(8):data(), 8, 1.0, 1.0, nil) end)   -- a host pointer as a device buffer
        vdnn.releaseStep()
        try(function() vd.call('vd_axpby', x, x, x, 8, 1.0, 1.0, nil) end)                   -- use after releaseStep
        try(function() vd.call('vd_axpby', ffi.cast('int32_t*', vdnn.devBytes(32)), nil, nil, 8, 1.0, 1.0, nil) end)   -- int32_t* for const float*
        try(function() vd.call('vd_no_such_entry_point') end)
        return out''')][0]
    assert 'overruns its allocation by 1 bytes' in errs[0]
    assert 'not device memory' in errs[1]
    assert 'FREED device memory' in errs[2]
    assert "cannot convert 'int32_t *' to 'const float *'" in errs[3]
    assert "missing declaration for symbol 'vd_no_such_entry_point'" in errs[4]
    # a batch with 0-based answer indices, or a token tensor that died before the upload, is refused at the boundary
    m = host.model()
    batch = SyntheticDataloader(p, seed=1).getTrainBatch(p)
    bad = dict(batch, answer_ind=batch['answer_ind'] - 1)
    with pytest.raises(LuaError, match='1-based'):
        host.invoke(m, 'forwardBackward', host.batch(bad))
    host.close()
