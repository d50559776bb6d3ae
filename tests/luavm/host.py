"""Glue between Python tests and a luavm state: VM construction (torch stub + ffi, optionally a DRY library for CPU runs) and value
conversion (dict / list / numpy <-> LuaTable / Tensor)."""
import numpy as np

from . import ffi as ffi_mod
from . import torch7
from .interp import LuaTable, LuaVM


def new_vm(search=(), lib_loader=None, seed=0, stdout=None):
    """a Lua state with `torch` (torch7.py) and `require 'ffi'` (ffi.py).  lib_loader(path, global) -> an object whose attributes
    are the C functions (default: ctypes.CDLL) -- tests on a box without a GPU pass a dry library (drylib.py)."""
    vm = LuaVM(search=search, stdout=stdout)
    torch7.install(vm, seed)
    ffi_mod.install(vm, lib_loader)
    return vm


def to_lua(vm, v, tensor_types=None, key=None):
    """Python -> Lua: dict -> table, list / tuple -> array table, numpy array -> Tensor (type from the dtype, or from
    tensor_types[key] -- e.g. {'ques_fwd': 'Long'} to hand a batch over in the tensor types the reference's dataloader produces)"""
    if isinstance(v, dict):
        t = LuaTable()
        for k, x in v.items():
            t.set(k, to_lua(vm, x, tensor_types, k))
        return t
    if isinstance(v, (list, tuple)):
        t = LuaTable()
        for i, x in enumerate(v):
            t.set(i + 1, to_lua(vm, x, tensor_types, key))
        return t
    if isinstance(v, np.ndarray):
        return vm.torch.tensor(v, (tensor_types or {}).get(key))
    if isinstance(v, (np.integer,)):
        return int(v)
    if isinstance(v, (np.floating,)):
        return float(v)
    return v


def to_py(v):
    """Lua -> Python: array-like tables -> list, other tables -> dict, Tensor -> numpy copy"""
    if isinstance(v, LuaTable):
        if v.arr and not v.hash:
            return [to_py(x) for x in v.arr]
        return {k: to_py(x) for k, x in v.items()}
    if isinstance(v, torch7.Tensor):
        return np.array(v.a, copy=True)
    return v
