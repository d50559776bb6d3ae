"""Test-side driver of the Lua host (lua/*.lua) inside luavm: builds a Lua state whose `ffi.load` is either the real
libvisdial_hip.so (GPU tests) or luavm.drylib.DryLib (CPU tests), hands batches over as Torch7 tensors of the types the reference's
dataloader produces (dataloader.lua:46-104: token tensors as read from HDF5 = IntTensor, `:long()` for answer / option indices,
FloatTensor image features), and exposes method calls on Lua objects."""
import os

import numpy as np

from conftest import ROOT
from luavm import new_vm, to_lua, to_py
from luavm.interp import LuaTable, call, index

LUA_DIR = os.path.join(ROOT, 'lua')
STUBS = os.path.join(ROOT, 'tests', 'lua_stubs')
# the tensor types a batch arrives in (the reference's dataloader on the CPU path, `-gpuid -1`)
BATCH_TYPES = {'ques_fwd': 'Int', 'hist': 'Int', 'options': 'Int', 'answer_in': 'Int', 'answer_out': 'Int', 'option_in': 'Int',
               'option_out': 'Int', 'answer_ind': 'Long', 'img_feat': 'Float'}


class PyDataloader(object):
    """a dataloader OBJECT for Lua (model:trainIteration(dataloader) calls dataloader:getTrainBatch(params)) over a Python one"""
    lua_type = 'table'

    def __init__(self, host, dl, params, types=None):
        self.host, self.dl, self.p, self.types = host, dl, params, types or BATCH_TYPES
        self.served = []
        self.fields = {}

    def lua_index(self, k):
        if k == 'getTrainBatch':
            def get(_self, _params=None, *_a):
                b = self.dl.getTrainBatch(self.p)
                self.served.append(b)
                return self.host.batch(b, self.types)
            return get
        if k == 'getTestBatch':
            def get_test(_self, start, _params, dtype, *_a):
                b, nxt = self.dl.getTestBatch(int(start), self.p, dtype)
                self.served.append(b)
                return self.host.batch(b, self.types), nxt
            return get_test
        if k == 'getIndexData':
            def get_index(_self, inds, _params, dtype, *_a):
                b = self.dl.getIndexData(np.asarray(to_py(inds)).reshape(-1).astype(np.int64), self.p, dtype)
                return self.host.batch(b, self.types)
            return get_index
        v = self.fields.get(k)
        if v is None and hasattr(self.dl, k):
            v = getattr(self.dl, k)
            if isinstance(v, (dict, list, np.ndarray)):
                v = to_lua(self.host.vm, v)
            self.fields[k] = v
        return v

    def lua_newindex(self, k, v):
        self.fields[k] = v


class LuaHost(object):
    def __init__(self, params, dry=False, env=None, seed=0):
        """params: the opts dict (visdial_amd.opts.derive).  dry=True: no GPU, luavm.drylib.DryLib behind ffi.load."""
        from visdial_amd import _lib
        self.p = dict(params)
        self.dry = None
        for k, v in (env or {}).items():
            os.environ[k] = v
        os.environ['VD_LIB_PATH'] = _lib.LIB_PATH
        loader = None
        if dry:
            from oracle import visdial_oracle as vo
            from luavm.drylib import DryLib

            def spec(enc, dec, cparams):
                q = dict(self.p)
                q.update({k: v for k, v in cparams.items() if k in q and k not in ('learningRate', 'lrDecayRate', 'minLRate', 'dropout')})
                return vo.param_spec(enc, dec, q)

            def loader(path, glob):
                self.dry = DryLib(self.vm.ffi, spec)
                return self.dry
        self.vm = new_vm(search=[LUA_DIR, STUBS], lib_loader=loader, seed=seed)
        self.vm.globals.set('runningLoss', 0)                   # train.lua:89 initialises the global before the loop

    # ---- values
    def params(self, **over):
        clean = {k: v for k, v in dict(self.p, **over).items() if isinstance(v, (int, float, str, bool))}
        return to_lua(self.vm, clean)

    def batch(self, b, types=None):
        return to_lua(self.vm, {k: v for k, v in b.items() if isinstance(v, np.ndarray)}, types or BATCH_TYPES)

    def dataloader(self, dl, types=None):
        return PyDataloader(self, dl, self.p, types)

    def tensor(self, a, tn=None):
        return self.vm.torch.tensor(np.asarray(a), tn)

    def invoke(self, obj, name, *args):
        """obj:name(...) -> list of results"""
        return call(index(obj, name), [obj] + list(args))

    def get(self, obj, *path):
        for k in path:
            obj = index(obj, k)
        return obj

    # ---- the two hosts
    def model(self, **over):
        """lua/model.lua: the model-level host (class Model of the reference)"""
        self.vm.dofile('model.lua')
        return first(call(self.vm.globals.get('Model'), [self.params(**over)]))

    def model_ops(self, **over):
        """lua/model_ops.lua: the operator-level host over plug-in files composed in Lua"""
        self.vm.dofile('model_ops.lua')
        return first(call(self.vm.globals.get('ModelOps'), [self.params(**over)]))

    # ---- ModelOps parameter access (vdnn.FlatParams)
    def ops_spec(self, m):
        fp = index(m, 'fp')
        order, size = to_py(index(fp, 'order')), index(fp, 'size')
        return [(n, int(size.get(n))) for n in order]

    def ops_set_parameters(self, m, named):
        flat = np.concatenate([np.asarray(named[n], np.float32).reshape(-1) for n, _ in self.ops_spec(m)])
        self.invoke(index(m, 'fp'), 'copyFrom', self.tensor(flat, 'Float'))

    def ops_get(self, m, which, shapes):
        flat = to_py(first(self.invoke(index(m, 'fp'), 'toFloat', which)))
        out, o = {}, 0
        for n, k in self.ops_spec(m):
            out[n] = flat[o:o + k].reshape(shapes[n])
            o += k
        return out

    def close(self):
        self.vm.close()


def first(vals):
    return vals[0] if vals else None
