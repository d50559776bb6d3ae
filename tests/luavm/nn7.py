"""Torch7 `nn` / `nngraph` / Element-Research `rnn` for luavm: the module classes the REFERENCE's Lua sources construct (model.lua,
encoders/*.lua, decoders/*.lua, model_utils/*.lua), implemented over numpy fp64 with Torch7's interfaces (forward / backward /
parameters / getParameters / clone with shared storages / training / evaluate, nngraph's `module(inputs)` node syntax and gModule).

Purpose (tests/golden/make_reference_goldens.py, BUILD CONTAINER ONLY -- /root/reference does not travel): execute the reference's own
files, unmodified, and record what they compute -- loss, gradients in getParameters() order, the update of optim_updates.lua's adam,
option ranks -- as fixtures that the repo's oracle and HIP path are then held to.  The third-party module SEMANTICS (what nn.SeqLSTM,
nn.LookupTableMaskZero, nn.MaskZero ... compute) are restated here from their published behaviour, written
independently of oracle/visdial_oracle.py; the WIRING -- which modules, in which order, with which sharing, what model.lua does with
their outputs -- is the reference's code running.  gModule orders its modules like nngraph does: post-order depth-first walk of the
graph from the output node over each node's inputs in declaration order (graph.Graph:topsort over the reversed forward graph).
"""
import numpy as np

from .interp import LuaError, LuaTable, call, first, index, setindex, type_name
from .torch7 import Tensor


def _tab(seq):
    t = LuaTable()
    for i, v in enumerate(seq):
        t.set(i + 1, v)
    return t


def _items(t):
    return [t.get(i + 1) for i in range(t.length())]


def fwd(mod, inp):
    return first(call(index(mod, 'forward'), [mod, inp]))


def bwd(mod, inp, gout):
    return first(call(index(mod, 'backward'), [mod, inp, gout]))


def bwd_split(mod, inp, gout):
    """nngraph calls updateGradInput and accGradParameters separately and uses the RETURN VALUE of updateGradInput (a Lua module may
    return a table there without storing it in self.gradInput: model_utils/MaskSoftMax.lua:45)"""
    gi = first(call(index(mod, 'updateGradInput'), [mod, inp, gout]))
    call(index(mod, 'accGradParameters'), [mod, inp, gout, 1])
    return gi


def tree_map(f, x):
    if isinstance(x, LuaTable):
        return _tab([tree_map(f, v) for v in _items(x)])
    return f(x)


def tree_add(a, b):
    if a is None:
        return b
    if b is None:
        return a
    if isinstance(a, LuaTable):
        return _tab([tree_add(x, y) for x, y in zip(_items(a), _items(b))])
    return a.new_like(a.a + b.a.reshape(a.a.shape))


def zeros_like(x):
    return tree_map(lambda t: t.new_like(np.zeros_like(t.a)), x)


class Module(object):
    lua_type = 'table'
    typename = 'nn.Module'

    def __init__(self, NN):
        self.NN = NN
        NN.created += 1
        self.created = NN.created                   # creation order (the Lua source's order of constructor calls)
        self.f = {'train': True}
        self.f['output'] = NN.T.f_Tensor()
        self.f['gradInput'] = NN.T.f_Tensor()

    # ---- Lua protocol
    def lua_index(self, k):
        if k in self.f:
            return self.f[k]
        if k == '__typename':
            return self.typename
        m = getattr(type(self), 'm_' + k, None) if k.__class__ is str else None
        return m

    def lua_newindex(self, k, v):
        if v is None:
            self.f.pop(k, None)
        else:
            self.f[k] = v

    def lua_tostring(self):
        return self.typename

    def lua_call(self, args):                       # nngraph: module(input nodes) -> node
        return self.NN.make_node(self, args)

    def arr(self, t, what='input'):
        if not isinstance(t, Tensor):
            raise LuaError('%s: tensor expected as %s, got %s' % (self.typename, what, type_name(t)))
        return t.a.astype(np.float64, copy=False)

    def out(self, a):
        return self.NN.T.tensor(np.asarray(a, dtype=np.float64), 'Double')

    # ---- nn.Module interface
    def m_forward(self, inp):
        self.f['output'] = self.updateOutput(inp)
        return self.f['output']

    def m_updateOutput(self, inp):
        return self.m_forward(inp)

    def m_backward(self, inp, gout, scale=1):
        self.f['gradInput'] = self.updateGradInput(inp, gout)
        return self.f['gradInput']

    def m_updateGradInput(self, inp, gout):
        return self.m_backward(inp, gout)

    def m_accGradParameters(self, *a):
        return None

    def updateOutput(self, inp):
        raise LuaError('%s: forward not implemented' % self.typename)

    def updateGradInput(self, inp, gout):
        raise LuaError('%s: backward not implemented' % self.typename)

    def params(self):
        """[(weight, gradWeight)] of THIS module, nn order (weight, bias)"""
        out = []
        for w, g in (('weight', 'gradWeight'), ('bias', 'gradBias')):
            if self.f.get(w) is not None:
                out.append((self.f[w], self.f[g]))
        return out

    def children(self):
        return []

    def all_params(self):
        out = list(self.params())
        for c in self.children():
            if isinstance(c, Module):
                out += c.all_params()
            else:                                   # a class written in Lua (model_utils/*.lua): no parameters of its own here
                r = call(index(c, 'parameters'), [c])
                if r and r[0] is not None:
                    out += list(zip(_items(r[0]), _items(r[1])))
        return out

    def m_parameters(self):
        ps = self.all_params()
        if not ps:
            return None
        return (_tab([p for p, _ in ps]), _tab([g for _, g in ps]))

    def m_getParameters(self):
        """nn.Module.flatten: one flat tensor per list, every DISTINCT storage once, at its first occurrence; the module tensors
        become views into it"""
        ps = self.all_params()
        flats = []
        for which in (0, 1):
            seen, total, order = {}, 0, []
            for pair in ps:
                t = pair[which]
                key = id(t.st)
                if key not in seen:
                    seen[key] = total
                    order.append(t)
                    total += t.st.base.size
            flat = self.NN.T.tensor(np.zeros(total), 'Double')
            for t in order:
                off = seen[id(t.st)]
                flat.a[off:off + t.st.base.size] = t.st.base
            # re-point every tensor (all aliases of a storage share the Tensor's storage object -> swap the storage's base)
            for t in order:
                off = seen[id(t.st)]
                view = flat.st.base[off:off + t.st.base.size]
                old_st = t.st
                for pair in ps:
                    for u in pair:
                        if u.st is old_st:
                            shape = u.a.shape
                            u.st = flat.st
                            u.a = view.reshape(shape)
            flats.append(flat)
        self.NN.last_flat_layout = [(id(p.st), p.a.shape) for p, _ in ps]
        return tuple(flats)

    def m_zeroGradParameters(self):
        for _, g in self.all_params():
            g.a[...] = 0
        return None

    def _mode(self, train):
        self.f['train'] = train
        for c in self.children():
            if isinstance(c, Module):
                c._mode(train)
            else:
                call(index(c, 'training' if train else 'evaluate'), [c])

    def m_training(self):
        self._mode(True)
        return self

    def m_evaluate(self):
        self._mode(False)
        return self

    def m_cuda(self):
        return self
    m_float = m_double = m_cuda

    def m_type(self, *_):
        return self

    def m_clone(self, *share):
        import copy
        new = copy.copy(self)
        new.f = dict(self.f)
        for k, v in list(new.f.items()):
            if isinstance(v, Tensor):
                new.f[k] = v if k in share else v.new_like(v.a.copy())
        self._clone_children(new, share)
        return new

    def _clone_children(self, new, share):
        pass

    def m_share(self, other, *names):
        for n in names:
            self.f[n] = other.f[n]
        return self

    def m_reset(self, *a):
        return None

    def m_get(self, i):
        return self.children()[int(i) - 1]

    def m_apply(self, fn):
        call(fn, [self])
        for c in self.children():
            if isinstance(c, Module):
                c.m_apply(fn)


class Container(Module):
    def __init__(self, NN):
        Module.__init__(self, NN)
        self.f['modules'] = LuaTable()

    def children(self):
        return _items(self.f['modules'])

    def m_add(self, m):
        self.f['modules'].set(self.f['modules'].length() + 1, m)
        return self

    def m_size(self):
        return self.f['modules'].length()

    def lua_len(self):
        return self.f['modules'].length()

    def _clone_children(self, new, share):
        new.f['modules'] = _tab([c.m_clone(*share) if isinstance(c, Module) else c for c in self.children()])


class Sequential(Container):
    typename = 'nn.Sequential'

    def updateOutput(self, inp):
        self.inputs = [inp]
        for m in self.children():
            inp = fwd(m, inp)
            self.inputs.append(inp)
        return inp

    def updateGradInput(self, inp, gout):
        ms = self.children()
        for i in range(len(ms) - 1, -1, -1):
            gout = bwd(ms[i], self.inputs[i] if i else inp, gout)
        return gout


class ConcatTable(Container):
    typename = 'nn.ConcatTable'

    def updateOutput(self, inp):
        return _tab([fwd(m, inp) for m in self.children()])

    def updateGradInput(self, inp, gout):
        g = None
        for m, go in zip(self.children(), _items(gout)):
            g = tree_add(g, bwd(m, inp, go))
        return g


class ParallelTable(Container):
    typename = 'nn.ParallelTable'

    def updateOutput(self, inp):
        return _tab([fwd(m, x) for m, x in zip(self.children(), _items(inp))])

    def updateGradInput(self, inp, gout):
        return _tab([bwd(m, x, g) for m, x, g in zip(self.children(), _items(inp), _items(gout))])


class Concat(Container):
    typename = 'nn.Concat'

    def __init__(self, NN, dim):
        Container.__init__(self, NN)
        self.dim = int(dim) - 1

    def updateOutput(self, inp):
        outs = [self.arr(fwd(m, inp), 'output') for m in self.children()]
        self.sizes = [o.shape[self.dim] for o in outs]
        return self.out(np.concatenate(outs, axis=self.dim))

    def updateGradInput(self, inp, gout):
        g, off, go = None, 0, self.arr(gout)
        for m, n in zip(self.children(), self.sizes):
            sl = [slice(None)] * go.ndim
            sl[self.dim] = slice(off, off + n)
            g = tree_add(g, bwd(m, inp, self.out(go[tuple(sl)])))
            off += n
        return g


class JoinTable(Module):
    typename = 'nn.JoinTable'

    def __init__(self, NN, dim, n_input_dims=None):
        Module.__init__(self, NN)
        self.dim, self.nid = int(dim), n_input_dims

    def axis(self, x):
        d = self.dim
        if d < 0:
            return x.ndim + d
        if self.nid is not None and x.ndim == int(self.nid) + 1:
            return d
        return d - 1

    def updateOutput(self, inp):
        xs = [self.arr(t) for t in _items(inp)]
        self.sizes = [x.shape[self.axis(x)] for x in xs]
        return self.out(np.concatenate(xs, axis=self.axis(xs[0])))

    def updateGradInput(self, inp, gout):
        go, out, off = self.arr(gout), [], 0
        ax = self.axis(go)
        for n in self.sizes:
            sl = [slice(None)] * go.ndim
            sl[ax] = slice(off, off + n)
            out.append(self.out(go[tuple(sl)]))
            off += n
        return _tab(out)


class SelectTable(Module):
    typename = 'nn.SelectTable'

    def __init__(self, NN, i):
        Module.__init__(self, NN)
        self.i = int(i)

    def pick(self, inp):
        n = inp.length()
        return self.i if self.i > 0 else n + self.i + 1

    def updateOutput(self, inp):
        return inp.get(self.pick(inp))

    def updateGradInput(self, inp, gout):
        k = self.pick(inp)
        return _tab([gout if j + 1 == k else zeros_like(x) for j, x in enumerate(_items(inp))])


class NarrowTable(Module):
    typename = 'nn.NarrowTable'

    def __init__(self, NN, offset, length=1):
        Module.__init__(self, NN)
        self.o, self.n = int(offset), int(length)

    def updateOutput(self, inp):
        return _tab(_items(inp)[self.o - 1:self.o - 1 + self.n])

    def updateGradInput(self, inp, gout):
        gs = _items(gout)
        return _tab([gs[j - (self.o - 1)] if self.o - 1 <= j < self.o - 1 + self.n else zeros_like(x) for j, x in enumerate(_items(inp))])


class Identity(Module):
    typename = 'nn.Identity'

    def updateOutput(self, inp):
        return inp

    def updateGradInput(self, inp, gout):
        return gout


class Linear(Module):
    typename = 'nn.Linear'

    def __init__(self, NN, n_in, n_out):
        Module.__init__(self, NN)
        n_in, n_out = int(n_in), int(n_out)
        stdv = 1.0 / np.sqrt(n_in)
        self.f['weight'] = self.out(NN.rng.uniform(-stdv, stdv, (n_out, n_in)))
        self.f['bias'] = self.out(NN.rng.uniform(-stdv, stdv, (n_out,)))
        self.f['gradWeight'] = self.out(np.zeros((n_out, n_in)))
        self.f['gradBias'] = self.out(np.zeros(n_out))

    def updateOutput(self, inp):
        x = self.arr(inp)
        return self.out(x @ self.f['weight'].a.T + self.f['bias'].a)

    def updateGradInput(self, inp, gout):
        x, g = self.arr(inp), self.arr(gout)
        x2, g2 = x.reshape(-1, x.shape[-1]), g.reshape(-1, g.shape[-1])
        self.f['gradWeight'].a[...] += g2.T @ x2
        self.f['gradBias'].a[...] += g2.sum(0)
        return self.out(g @ self.f['weight'].a)

    def m_reset(self, stdv=None):
        n_out, n_in = self.f['weight'].a.shape
        stdv = stdv * np.sqrt(3) if stdv is not None else 1.0 / np.sqrt(n_in)
        self.f['weight'].a[...] = self.NN.rng.uniform(-stdv, stdv, (n_out, n_in))
        self.f['bias'].a[...] = self.NN.rng.uniform(-stdv, stdv, (n_out,))
        return self


class Tanh(Module):
    typename = 'nn.Tanh'

    def updateOutput(self, inp):
        self.y = np.tanh(self.arr(inp))
        return self.out(self.y)

    def updateGradInput(self, inp, gout):
        return self.out(self.arr(gout) * (1 - self.y * self.y))


class SoftMax(Module):
    typename = 'nn.SoftMax'

    def updateOutput(self, inp):
        x = self.arr(inp)
        e = np.exp(x - x.max(-1, keepdims=True))
        self.y = e / e.sum(-1, keepdims=True)
        return self.out(self.y)

    def updateGradInput(self, inp, gout):
        g = self.arr(gout)
        return self.out(self.y * (g - (g * self.y).sum(-1, keepdims=True)))


class LogSoftMax(Module):
    typename = 'nn.LogSoftMax'

    def updateOutput(self, inp):
        x = self.arr(inp)
        m = x.max(-1, keepdims=True)
        self.y = x - m - np.log(np.exp(x - m).sum(-1, keepdims=True))
        return self.out(self.y)

    def updateGradInput(self, inp, gout):
        g = self.arr(gout)
        return self.out(g - np.exp(self.y) * g.sum(-1, keepdims=True))


class Dropout(Module):
    typename = 'nn.Dropout'

    def __init__(self, NN, p=0.5):
        Module.__init__(self, NN)
        self.p = float(p)

    def updateOutput(self, inp):
        x = self.arr(inp)
        if not self.f['train'] or self.p <= 0:
            self.noise = None
            return self.out(x)
        self.noise = (self.NN.rng.uniform(size=x.shape) > self.p).astype(np.float64)     # keep-mask
        self.NN.dropout_log.append((self, self.noise))        # (module, the keep-mask of THIS call), in forward order
        return self.out(x * self.noise / (1 - self.p))

    def updateGradInput(self, inp, gout):
        g = self.arr(gout)
        return self.out(g if self.noise is None else g * self.noise / (1 - self.p))


class LookupTableMaskZero(Module):
    typename = 'nn.LookupTableMaskZero'

    def __init__(self, NN, V, E):
        Module.__init__(self, NN)
        V, E = int(V), int(E)
        self.f['weight'] = self.out(NN.rng.normal(size=(V + 1, E)))
        self.f['gradWeight'] = self.out(np.zeros((V + 1, E)))

    def updateOutput(self, inp):
        self.f['weight'].a[0] = 0                            # the pad row is re-zeroed on every forward
        self.idx = np.asarray(inp.a, dtype=np.int64)
        return self.out(self.f['weight'].a[self.idx])        # index + 1 in 1-based terms = row `index` 0-based

    def updateGradInput(self, inp, gout):
        g = self.arr(gout)
        np.add.at(self.f['gradWeight'].a, self.idx.reshape(-1), g.reshape(-1, g.shape[-1]))
        return inp.new_like(np.zeros_like(inp.a))            # nn.LookupTable: zeros shaped like the index tensor


def _sig(x):
    return 1.0 / (1.0 + np.exp(-x))


class SeqLSTM(Module):
    """Element-Research rnn nn.SeqLSTM: weight [(D + H) x 4H] = [Wx; Wh], gate columns i, f, o, g; bias [4H], forget
    slice = 1; time-major unless .batchfirst; maskZero(): a row whose input vector is all-zero at step t gets h = c = gates = 0"""
    typename = 'nn.SeqLSTM'

    def __init__(self, NN, D, H):
        Module.__init__(self, NN)
        D, H = int(D), int(H)
        self.D, self.H = D, H
        std = 1.0 / np.sqrt(H + D)
        self.f['weight'] = self.out(NN.rng.normal(size=(D + H, 4 * H)) * std)
        b = np.zeros(4 * H)
        b[H:2 * H] = 1
        self.f['bias'] = self.out(b)
        self.f['gradWeight'] = self.out(np.zeros((D + H, 4 * H)))
        self.f['gradBias'] = self.out(np.zeros(4 * H))
        self.f['maskzero'] = False

    def m_maskZero(self):
        self.f['maskzero'] = True
        return self

    def updateOutput(self, inp):
        x = self.arr(inp)
        if self.f.get('batchfirst'):
            x = x.transpose(1, 0, 2)
        T, N, D = x.shape
        H, W, b = self.H, self.f['weight'].a, self.f['bias'].a
        h0, c0 = self.f.pop('userPrevOutput', None), self.f.pop('userPrevCell', None)
        hp = self.arr(h0).copy() if h0 is not None else np.zeros((N, H))
        cp = self.arr(c0).copy() if c0 is not None else np.zeros((N, H))
        self.h0, self.c0, self.had0 = hp, cp, (h0 is not None, c0 is not None)
        hs, cs, gs = np.zeros((T, N, H)), np.zeros((T, N, H)), np.zeros((T, N, 4 * H))
        self.x = x
        for t in range(T):
            a = b + x[t] @ W[:D] + hp @ W[D:]
            g = np.concatenate([_sig(a[:, :3 * H]), np.tanh(a[:, 3 * H:])], 1)
            c = g[:, H:2 * H] * cp + g[:, :H] * g[:, 3 * H:]
            h = g[:, 2 * H:3 * H] * np.tanh(c)
            if self.f['maskzero']:
                dead = ~x[t].any(1)
                g[dead], c[dead], h[dead] = 0, 0, 0
            hs[t], cs[t], gs[t] = h, c, g
            hp, cp = h, c
        self.hs, self.cs, self.gs = hs, cs, gs
        self.f['cell'] = self.out(cs)
        o = hs.transpose(1, 0, 2) if self.f.get('batchfirst') else hs
        return self.out(o)

    def updateGradInput(self, inp, gout):
        go = self.arr(gout)
        if self.f.get('batchfirst'):
            go = go.transpose(1, 0, 2)
        T, N, H, D = self.x.shape[0], self.x.shape[1], self.H, self.D
        W = self.f['weight'].a
        gpo, ngc = self.f.pop('gradPrevOutput', None), self.f.pop('userNextGradCell', None)
        dh_next = self.arr(gpo).copy() if gpo is not None else np.zeros((N, H))
        dc = self.arr(ngc).copy() if ngc is not None else np.zeros((N, H))
        dx = np.zeros_like(self.x)
        for t in range(T - 1, -1, -1):
            g, c = self.gs[t], self.cs[t]
            cprev = self.cs[t - 1] if t else self.c0
            hprev = self.hs[t - 1] if t else self.h0
            i, f_, o, gg = g[:, :H], g[:, H:2 * H], g[:, 2 * H:3 * H], g[:, 3 * H:]
            dh = go[t] + dh_next
            tc = np.tanh(c)
            dc = dc + o * (1 - tc * tc) * dh
            da = np.concatenate([i * (1 - i) * gg * dc, f_ * (1 - f_) * cprev * dc, o * (1 - o) * tc * dh, i * (1 - gg * gg) * dc], 1)
            dx[t] = da @ W[:D].T
            self.f['gradWeight'].a[:D] += self.x[t].T @ da
            self.f['gradWeight'].a[D:] += hprev.T @ da
            self.f['gradBias'].a[...] += da.sum(0)
            dh_next = da @ W[D:].T
            dc = dc * f_
        self.f['userGradPrevOutput'] = self.out(dh_next)
        self.f['userGradPrevCell'] = self.out(dc)
        if self.f.get('batchfirst'):
            dx = dx.transpose(1, 0, 2)
        return self.out(dx)


class Select(Module):
    typename = 'nn.Select'

    def __init__(self, NN, dim, idx):
        Module.__init__(self, NN)
        self.dim, self.idx = int(dim), int(idx)

    def sl(self, x):
        d = self.dim - 1 if self.dim > 0 else x.ndim + self.dim
        i = self.idx - 1 if self.idx > 0 else x.shape[d] + self.idx
        return (slice(None),) * d + (i,)

    def updateOutput(self, inp):
        x = self.arr(inp)
        return self.out(x[self.sl(x)])

    def updateGradInput(self, inp, gout):
        x = self.arr(inp)
        g = np.zeros_like(x)
        g[self.sl(x)] = self.arr(gout)
        return self.out(g)


class View(Module):
    typename = 'nn.View'

    def __init__(self, NN, *sizes):
        Module.__init__(self, NN)
        self.sizes, self.nid = [int(s) for s in sizes], None

    def m_setNumInputDims(self, n):
        self.nid = int(n)
        return self

    def target_shape(self, x):
        """nn.View's batchsize(): the trailing numInputDims (default: all) dimensions are re-viewed as `sizes`; what is left over is a
        leading batch dimension -- unless `sizes` holds a -1, which absorbs it"""
        ind = x.ndim
        maxdim = self.nid if self.nid is not None else ind
        ine = int(np.prod(x.shape[ind - maxdim:])) if maxdim > 0 else 1
        ne = int(np.prod([s for s in self.sizes if s > 0])) if any(s > 0 for s in self.sizes) else 1
        if ine % ne != 0:
            raise LuaError('nn.View: input view (%s) and desired view (%s) do not match' % (x.shape, self.sizes))
        bsz = ine // ne
        if -1 in self.sizes:
            bsz = 1
        for d in x.shape[:ind - maxdim]:
            bsz *= d
        if bsz == 1 and (self.nid is None or ind <= self.nid):
            return list(self.sizes)
        return [bsz] + list(self.sizes)

    def updateOutput(self, inp):
        x = self.arr(inp)
        return self.out(x.reshape(self.target_shape(x)))

    def updateGradInput(self, inp, gout):
        return self.out(self.arr(gout).reshape(self.arr(inp).shape))


class Reshape(Module):
    typename = 'nn.Reshape'

    def __init__(self, NN, *args):
        Module.__init__(self, NN)
        args = list(args)
        self.batch = None
        if args and args[-1].__class__ is bool:
            self.batch = args.pop()
        self.sizes = [int(a) for a in args]

    def updateOutput(self, inp):
        x = self.arr(inp)
        n = int(np.prod(self.sizes))
        if self.batch or (self.batch is None and x.size != n):
            return self.out(x.reshape([x.shape[0]] + self.sizes))
        return self.out(x.reshape(self.sizes))

    def updateGradInput(self, inp, gout):
        return self.out(self.arr(gout).reshape(self.arr(inp).shape))


class Transpose(Module):
    typename = 'nn.Transpose'

    def __init__(self, NN, *perms):
        Module.__init__(self, NN)
        self.perms = [(int(p.get(1)) - 1, int(p.get(2)) - 1) for p in perms]

    def updateOutput(self, inp):
        x = self.arr(inp)
        for a, b in self.perms:
            x = np.swapaxes(x, a, b)
        return self.out(x)

    def updateGradInput(self, inp, gout):
        g = self.arr(gout)
        for a, b in reversed(self.perms):
            g = np.swapaxes(g, a, b)
        return self.out(g)


class Replicate(Module):
    typename = 'nn.Replicate'

    def __init__(self, NN, n, dim=1, ndim=None):
        Module.__init__(self, NN)
        self.n, self.dim = int(n), int(dim) - 1

    def updateOutput(self, inp):
        x = self.arr(inp)
        return self.out(np.repeat(np.expand_dims(x, self.dim), self.n, axis=self.dim))

    def updateGradInput(self, inp, gout):
        return self.out(self.arr(gout).sum(self.dim))


class Squeeze(Module):
    typename = 'nn.Squeeze'

    def updateOutput(self, inp):
        return self.out(np.squeeze(self.arr(inp)))

    def updateGradInput(self, inp, gout):
        return self.out(self.arr(gout).reshape(self.arr(inp).shape))


class Sum(Module):
    typename = 'nn.Sum'

    def __init__(self, NN, dim=1, *a):
        Module.__init__(self, NN)
        self.dim = int(dim)

    def ax(self, x):
        return self.dim - 1 if self.dim > 0 else x.ndim + self.dim

    def updateOutput(self, inp):
        x = self.arr(inp)
        return self.out(x.sum(self.ax(x)))

    def updateGradInput(self, inp, gout):
        x = self.arr(inp)
        return self.out(np.broadcast_to(np.expand_dims(self.arr(gout), self.ax(x)), x.shape).copy())


class MM(Module):
    typename = 'nn.MM'

    def __init__(self, NN, ta=False, tb=False):
        Module.__init__(self, NN)
        self.ta, self.tb = bool(ta), bool(tb)

    def updateOutput(self, inp):
        a, b = (self.arr(t) for t in _items(inp))
        A = np.swapaxes(a, -1, -2) if self.ta else a
        B = np.swapaxes(b, -1, -2) if self.tb else b
        return self.out(A @ B)

    def updateGradInput(self, inp, gout):
        a, b = (self.arr(t) for t in _items(inp))
        g = self.arr(gout)
        A = np.swapaxes(a, -1, -2) if self.ta else a
        B = np.swapaxes(b, -1, -2) if self.tb else b
        dA, dB = g @ np.swapaxes(B, -1, -2), np.swapaxes(A, -1, -2) @ g
        if self.ta:
            dA = np.swapaxes(dA, -1, -2)
        if self.tb:
            dB = np.swapaxes(dB, -1, -2)
        return _tab([self.out(dA), self.out(dB)])


class CAddTable(Module):
    typename = 'nn.CAddTable'

    def updateOutput(self, inp):
        xs = [self.arr(t) for t in _items(inp)]
        return self.out(sum(xs[1:], xs[0].copy()))

    def updateGradInput(self, inp, gout):
        return _tab([self.out(self.arr(gout).reshape(self.arr(t).shape)) for t in _items(inp)])


class CMulTable(Module):
    typename = 'nn.CMulTable'

    def updateOutput(self, inp):
        xs = [self.arr(t) for t in _items(inp)]
        y = xs[0].copy()
        for x in xs[1:]:
            y = y * x
        return self.out(y)

    def updateGradInput(self, inp, gout):
        xs, g = [self.arr(t) for t in _items(inp)], self.arr(gout)
        out = []
        for i in range(len(xs)):
            p = g.copy()
            for j, x in enumerate(xs):
                if j != i:
                    p = p * x
            out.append(self.out(p))
        return _tab(out)


class MaskZero(Container):
    """rnn nn.MaskZero(module, nInputDim): rows of the OUTPUT whose INPUT row is all-zero are zeroed (and their gradient)"""
    typename = 'nn.MaskZero'

    def __init__(self, NN, module, n_input_dim=1):
        Container.__init__(self, NN)
        self.m_add(module)

    def updateOutput(self, inp):
        x = self.arr(inp)
        self.dead = ~x.reshape(-1, x.shape[-1]).any(1)
        y = self.arr(fwd(self.children()[0], inp), 'output').copy()
        y.reshape(-1, y.shape[-1])[self.dead] = 0
        return self.out(y)

    def updateGradInput(self, inp, gout):
        g = self.arr(gout).copy()
        g.reshape(-1, g.shape[-1])[self.dead] = 0
        gi = self.arr(bwd(self.children()[0], inp, self.out(g)), 'gradInput').copy()
        gi.reshape(-1, gi.shape[-1])[self.dead] = 0
        return self.out(gi)


class Sequencer(Container):
    """rnn nn.Sequencer(module) over a [T x N x ...] tensor: the module at every time step; for the stateless modules it wraps here
    (MaskZero(Linear), MaskZero(LogSoftMax)) that is the module applied to the [T*N x ...] rows at once"""
    typename = 'nn.Sequencer'

    def __init__(self, NN, module):
        Container.__init__(self, NN)
        self.m_add(module)

    def updateOutput(self, inp):
        x = self.arr(inp)
        self.shape = x.shape
        y = self.arr(fwd(self.children()[0], self.out(x.reshape(-1, x.shape[-1]))), 'output')
        return self.out(y.reshape(x.shape[:-1] + (y.shape[-1],)))

    def updateGradInput(self, inp, gout):
        x, g = self.arr(inp), self.arr(gout)
        gi = self.arr(bwd(self.children()[0], self.out(x.reshape(-1, x.shape[-1])), self.out(g.reshape(-1, g.shape[-1]))), 'gradInput')
        return self.out(gi.reshape(x.shape))


# ------------------------------------------------------------------------------------------------------------------- criteria
class Criterion(Module):
    def m_forward(self, inp, target):
        self.f['output'] = self.loss(inp, target)
        return self.f['output']

    def m_backward(self, inp, target):
        self.f['gradInput'] = self.grad(inp, target)
        return self.f['gradInput']


class ClassNLLCriterion(Criterion):
    typename = 'nn.ClassNLLCriterion'

    def __init__(self, NN, *a):
        Module.__init__(self, NN)
        self.f['sizeAverage'] = True

    def loss(self, inp, target):
        x, t = self.arr(inp), np.asarray(target.a, np.int64).reshape(-1) - 1
        v = -x[np.arange(x.shape[0]), t].sum()
        return float(v / x.shape[0]) if self.f['sizeAverage'] else float(v)

    def grad(self, inp, target):
        x, t = self.arr(inp), np.asarray(target.a, np.int64).reshape(-1) - 1
        g = np.zeros_like(x)
        g[np.arange(x.shape[0]), t] = -1.0 / x.shape[0] if self.f['sizeAverage'] else -1.0
        return self.out(g)


class CrossEntropyCriterion(Criterion):
    typename = 'nn.CrossEntropyCriterion'

    def __init__(self, NN, *a):
        Module.__init__(self, NN)
        self.lsm, self.nll = LogSoftMax(NN), ClassNLLCriterion(NN)

    def loss(self, inp, target):
        self.lp = fwd(self.lsm, inp)
        return self.nll.loss(self.lp, target)

    def grad(self, inp, target):
        return bwd(self.lsm, inp, self.nll.grad(self.lp, target))


class MaskZeroCriterion(Criterion):
    """rnn: rows of the input that are all-zero (the log-prob rows nn.MaskZero zeroed at padded steps) are skipped"""
    typename = 'nn.MaskZeroCriterion'

    def __init__(self, NN, crit, n_input_dim=1):
        Module.__init__(self, NN)
        self.crit = crit

    def live(self, inp):
        x = self.arr(inp)
        return x.reshape(-1, x.shape[-1]).any(1)

    def loss(self, inp, target):
        x, keep = self.arr(inp), self.live(inp)
        if not keep.any():
            return 0.0
        t = np.asarray(target.a).reshape(-1)
        return first(call(index(self.crit, 'forward'), [self.crit, self.out(x[keep]), self.NN.T.tensor(t[keep], 'Long')]))

    def grad(self, inp, target):
        x, keep = self.arr(inp), self.live(inp)
        g = np.zeros_like(x)
        if keep.any():
            t = np.asarray(target.a).reshape(-1)
            g[keep] = first(call(index(self.crit, 'backward'), [self.crit, self.out(x[keep]), self.NN.T.tensor(t[keep], 'Long')])).a
        return self.out(g)


class SequencerCriterion(Criterion):
    """rnn: the criterion at every time step of [T x N x V] / [T x N], losses summed"""
    typename = 'nn.SequencerCriterion'

    def __init__(self, NN, crit):
        Module.__init__(self, NN)
        self.crit = crit

    def loss(self, inp, target):
        x, t = self.arr(inp), np.asarray(target.a)
        total = 0.0
        for s in range(x.shape[0]):
            total += first(call(index(self.crit, 'forward'), [self.crit, self.out(x[s]), self.NN.T.tensor(t[s], 'Long')]))
        return total

    def grad(self, inp, target):
        x, t = self.arr(inp), np.asarray(target.a)
        g = np.zeros_like(x)
        for s in range(x.shape[0]):
            g[s] = first(call(index(self.crit, 'backward'), [self.crit, self.out(x[s]), self.NN.T.tensor(t[s], 'Long')])).a
        return self.out(g)


# ------------------------------------------------------------------------------------------------------------------- nngraph
class Node(object):
    """nngraph.Node: children = the nodes this one takes its inputs from, in the order they were passed (the graph is built backwards)"""
    lua_type = 'table'

    def __init__(self, NN, module=None, kind='module', index_=None):
        self.NN, self.module, self.kind, self.sel = NN, module, kind, index_
        self.children = []
        self.uid = NN.next_node_id()
        self.f = {}

    def lua_index(self, k):
        if k == 'split':
            return Node.m_split
        if k == 'annotate':
            return lambda s, *a: s
        if k == '__typename':
            return 'nngraph.Node'
        return self.f.get(k)

    def lua_newindex(self, k, v):
        self.f[k] = v

    def m_split(self, n):
        sp = Node(self.NN, kind='split')
        sp.children.append(self)
        outs = []
        for i in range(int(n)):
            s = Node(self.NN, kind='select', index_=i)
            s.children.append(sp)
            outs.append(s)
        return tuple(outs)


class gModule(Container):
    typename = 'nn.gModule'

    def __init__(self, NN, inputs, outputs):
        Container.__init__(self, NN)
        self.inputs, self.outputs = _items(inputs), _items(outputs)
        self.outnode = Node(NN, kind='out')
        self.outnode.children = list(self.outputs)
        self.innode = Node(NN, kind='in')
        if len(self.inputs) == 1:
            self.inputs[0].children.append(self.innode)
        else:
            sp = Node(NN, kind='split')
            sp.children.append(self.innode)
            for i, n in enumerate(self.inputs):
                if n.children:
                    raise LuaError('nngraph: an input should have no inputs')
                s = Node(NN, kind='select', index_=i)
                s.children.append(sp)
                n.children.append(s)
        # forward order = graph.Graph:topsort() of the forward graph = depth-first POST-order from the output node over each node's
        # inputs in declaration order (every node after all of its inputs)
        order, seen = [], set()

        def dfs(n):
            if n.uid in seen:
                return
            seen.add(n.uid)
            for c in n.children:
                dfs(c)
            order.append(n)
        import sys
        sys.setrecursionlimit(max(sys.getrecursionlimit(), 100000))
        dfs(self.outnode)
        self.order = order
        for n in order:
            if n.module is not None:
                self.m_add(n.module)
        self.f['forwardnodes'] = _tab(order)

    def node_inputs(self, n, vals):
        xs = [vals[c.uid] for c in n.children]
        return xs[0] if len(xs) == 1 else _tab(xs)

    def updateOutput(self, inp):
        vals = {}
        for n in self.order:
            if n.kind == 'in':
                vals[n.uid] = inp
            elif n.kind == 'split':
                vals[n.uid] = vals[n.children[0].uid]
            elif n.kind == 'select':
                vals[n.uid] = vals[n.children[0].uid].get(n.sel + 1)
            elif n.kind == 'out':
                vals[n.uid] = self.node_inputs(n, vals)
            else:
                vals[n.uid] = fwd(n.module, self.node_inputs(n, vals))
        self.vals = vals
        return vals[self.outnode.uid]

    def updateGradInput(self, inp, gout):
        grads = {self.outnode.uid: gout}
        for n in reversed(self.order):
            g = grads.get(n.uid)
            if n.kind == 'in':
                continue
            if n.kind == 'out':
                gi = g
            elif n.kind == 'split':
                # g: {select index -> gradient}; missing entries are zeros
                src = self.vals[n.children[0].uid]
                gi = _tab([g.get(i) if (g is not None and i in g) else zeros_like(src.get(i + 1)) for i in range(src.length())]) \
                    if isinstance(g, dict) else g
            elif n.kind == 'select':
                par = n.children[0]
                d = grads.setdefault(par.uid, {})
                d[n.sel] = tree_add(d.get(n.sel), g)
                continue
            else:
                if g is None:
                    g = zeros_like(self.vals[n.uid])
                gi = bwd_split(n.module, self.node_inputs(n, self.vals), g)
            if len(n.children) == 1:
                parts = [gi]
            else:
                parts = _items(gi)
            for c, p in zip(n.children, parts):
                if c.kind == 'in':
                    grads[c.uid] = tree_add(grads.get(c.uid), p) if not isinstance(p, dict) else p
                elif c.kind in ('split',):
                    grads[c.uid] = p
                else:
                    grads[c.uid] = tree_add(grads.get(c.uid), p)
        return grads.get(self.innode.uid)


CLASSES = [Sequential, ConcatTable, ParallelTable, Concat, JoinTable, SelectTable, NarrowTable, Identity, Linear, Tanh, SoftMax, LogSoftMax,
           Dropout, LookupTableMaskZero, SeqLSTM, Select, View, Reshape, Transpose, Replicate, Squeeze, Sum, MM, CAddTable, CMulTable, MaskZero,
           Sequencer, ClassNLLCriterion, CrossEntropyCriterion, MaskZeroCriterion, SequencerCriterion, gModule]

# nn.Module in LUA: the base class that classes written in Lua derive from (model_utils/MaskSoftMax.lua, MaskTime.lua, ...)
MODULE_LUA = r'''
local Module = torch.class('nn.Module')
function Module:__init() self.gradInput = torch.Tensor(); self.output = torch.Tensor(); self.train = true end
function Module:parameters() if self.weight and self.bias then return {self.weight, self.bias}, {self.gradWeight, self.gradBias} elseif self.weight then return {self.weight}, {self.gradWeight} end end
function Module:updateOutput(input) return self.output end
function Module:forward(input) return self:updateOutput(input) end
function Module:backward(input, gradOutput, scale) scale = scale or 1; self:updateGradInput(input, gradOutput); self:accGradParameters(input, gradOutput, scale); return self.gradInput end
function Module:updateGradInput(input, gradOutput) return self.gradInput end
function Module:accGradParameters(input, gradOutput, scale) end
function Module:zeroGradParameters() end
function Module:training() self.train = true end
function Module:evaluate() self.train = false end
function Module:cuda() return self end
function Module:float() return self end
function Module:double() return self end
function Module:type() return self end
function Module:__call__(...) return nngraph.__make_node(self, ...) end
'''


class NN(object):
    def __init__(self, vm, seed=0):
        self.vm, self.T = vm, vm.torch
        self.rng = np.random.RandomState(seed)
        self.dropout_log = []
        self.created = 0
        self._nid = 0
        self.last_flat_layout = None
        nn = self.module = LuaTable()
        vm.globals.set('nn', nn)                       # torch.class('nn.Module') below puts its constructor into this table
        g = LuaTable()
        vm.globals.set('nngraph', g)
        g.set('__make_node', lambda mod, *args: self.make_node(mod, list(args)))
        g.set('Node', lambda *a: Node(self, kind='module'))
        vm.dostring(MODULE_LUA, name='=nn.Module')
        for cls in CLASSES:
            nn.set(cls.typename.split('.')[1], self._ctor(cls))
        fast = LuaTable()
        nn.set('FastLSTM', fast)
        for name in ('VerticalConvolution', 'TemporalConvolution', 'SpatialConvolutionMM', 'SpatialConvolution', 'LateralConvolution',
                     'HorizontalConvolution'):
            nn.set(name, LuaTable())
        for name in ('nn', 'nngraph', 'rnn', 'cunn', 'cutorch', 'cudnn'):
            vm.loaded.set(name, nn if name != 'nngraph' else g)
        vm.nn = self

    def _ctor(self, cls):
        NN_ = self

        class Ctor(object):
            lua_type = 'table'

            def lua_call(self, args):
                return cls(NN_, *args)

            def lua_index(self, k):
                return getattr(cls, 'm_' + k, None) if k.__class__ is str else None

            def lua_newindex(self, k, v):
                setattr(cls, 'cls_' + str(k), v)
        return Ctor()

    def next_node_id(self):
        self._nid += 1
        return self._nid

    def make_node(self, module, args):
        if len(args) > 1:
            raise LuaError('nngraph: use {input1, input2} to pass multiple inputs')
        node = Node(self, module=module)
        inp = args[0] if args else None
        if inp is None:
            return node
        ins = _items(inp) if isinstance(inp, LuaTable) else [inp]
        for d in ins:
            if not isinstance(d, Node):
                raise LuaError('nngraph: what is this in the input? ' + type_name(d))
            node.children.append(d)
        return node


def install(vm, seed=0):
    return NN(vm, seed)
