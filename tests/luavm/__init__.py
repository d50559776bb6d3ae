"""luavm -- a Lua 5.1 interpreter with a LuaJIT-style `ffi` (over ctypes) and a Torch7 tensor stub (over numpy), in Python.

TEST INFRASTRUCTURE.  Neither the build container nor the GPU box has Lua, LuaJIT or Torch7, so the Lua host of this repo
(lua/model.lua, lua/model_ops.lua, lua/vdnn.lua, lua/visdial_ffi.lua and the plug-in files under lua/encoders, lua/decoders)
could only ever be parsed.  This package EXECUTES it: the Lua source is compiled to Python closures (parse.py, interp.py), its
`require 'ffi'` is answered by ffi.py -- a C declaration parser + cdata objects with LuaJIT's conversion rules, calling the real
libvisdial_hip.so through ctypes -- and its `torch` global by torch7.py (1-based, strided, typed tensors with Torch7's method
names).  Nothing under visdial_amd/ imports this package; nothing in it comes from the reference.

    from luavm import new_vm
    vm = new_vm(search=[ROOT + '/lua'])
    Model = vm.dofile('model.lua')
"""
from .interp import LuaError, LuaTable, LuaVM  # noqa: F401
from .host import new_vm, to_lua, to_py  # noqa: F401
