"""A DRY stand-in for libvisdial_hip.so, for running the Lua host on a box without a GPU (`-m "not gpu"`): same symbols, no arithmetic.

What it does do: a host-memory "device heap" with exact allocation sizes, so every vd_memset / vd_memcpy_* is bounds-checked and
every pointer argument of every other entry point must be NULL or lie inside a live allocation (a host pointer handed to a kernel,
a freed buffer, an int passed where a pointer belongs -> an error, not a crash); the model-level object keeps named parameter
tensors (layout from the oracle's param_spec -- this file is test infrastructure and may import it), a learning rate with the
library's decay rule, and returns deterministic stand-in numbers for loss / ranks / log-probabilities.  Every call is recorded in
`.calls` as (name, raw args).  The point is control flow: each Lua file executes end to end on the CPU and runtime errors of the
nil-index / wrong-arity / wrong-pointer-type kind surface here; numbers are checked on the GPU against the real library.
"""
import ctypes

import numpy as np

from .ffi import Ptr, Struct, Num, _struct_get
from .interp import LuaError


class DryError(LuaError):
    pass


class DryLib(object):
    def __init__(self, ffi, param_spec=None):
        self.ffi = ffi
        self.heap = {}                 # addr -> (buffer, nbytes)
        self.freed = set()
        self.calls = []
        self.models = {}
        self.param_spec = param_spec   # (encoder, decoder, params dict) -> [(name, shape, kind)]
        self.last_error = ctypes.create_string_buffer(b'', 512)
        self.comm = None
        self.bytes_allocated = 0
        self.peak_bytes = 0

    # ---- symbol resolution: explicit methods first, then a checked no-op
    def __getattr__(self, name):
        if name.startswith('__'):
            raise AttributeError(name)
        ct = self.ffi.decls.get(name)
        if ct is None:
            raise AttributeError(name)
        impl = self.__class__.__dict__.get('c_' + name)

        def fn(*raw):
            self.calls.append((name, raw))
            if impl is not None:
                return impl(self, *raw)
            self.check_pointers(name, ct, raw)
            return 0
        fn.__name__ = name
        return fn

    def fail(self, msg):
        self.last_error.value = msg.encode()[:500]
        return -1

    def find(self, addr):
        for base, (buf, n) in self.heap.items():
            if base <= addr < base + n:
                return base, n
        return None

    def check_pointers(self, name, ct, raw):
        for i, (p, v) in enumerate(zip(ct.params, raw)):
            if isinstance(p, Ptr) and v:
                if name.startswith('vd_model_'):
                    if i == 0 and v not in self.models:
                        raise DryError('%s: argument #1 is not a live vd_model handle' % name)
                    continue                                      # the model-level calls take HOST pointers
                if self.find(v) is None:
                    state = 'FREED device memory' if any(b <= v < b + n for b, n in self.freed) else 'not device memory (a host pointer?)'
                    raise DryError('%s: pointer argument #%d (%s) = 0x%x is %s' % (name, i + 1, p, v, state))

    def span(self, name, addr, nbytes, what):
        hit = self.find(addr)
        if hit is None:
            raise DryError('%s: %s 0x%x is not inside a live device allocation' % (name, what, addr))
        base, n = hit
        if addr + nbytes > base + n:
            raise DryError('%s: %s overruns its allocation by %d bytes (%d requested at offset %d of %d)'
                           % (name, what, addr + nbytes - base - n, nbytes, addr - base, n))

    # ---- core
    def c_vd_last_error(self):
        return ctypes.addressof(self.last_error)

    def c_vd_abi_version(self):
        import os
        import re
        hdr = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..', 'include', 'visdial_hip.h')).read()
        return int(re.search(r'#define\s+VD_ABI_VERSION\s+(\d+)', hdr).group(1))

    def c_vd_device_count(self, out):
        ctypes.c_int.from_address(out).value = 1
        return 0

    def c_vd_set_device(self, dev):
        return 0 if dev == 0 else self.fail('dry: device %d does not exist' % dev)

    def c_vd_malloc(self, out, nbytes):
        if nbytes <= 0:
            return self.fail('vd_malloc: bytes <= 0')
        buf = ctypes.create_string_buffer(int(nbytes))
        ctypes.memset(buf, 0xA5, int(nbytes))                    # fresh device memory is NOT zero
        addr = ctypes.addressof(buf)
        self.heap[addr] = (buf, int(nbytes))
        self.bytes_allocated += int(nbytes)
        self.peak_bytes = max(self.peak_bytes, self.bytes_allocated)
        ctypes.c_void_p.from_address(out).value = addr
        return 0

    def c_vd_free(self, ptr):
        if not ptr:
            return 0
        if ptr not in self.heap:
            raise DryError('vd_free: 0x%x is not the start of a live allocation (double free?)' % ptr)
        buf, n = self.heap.pop(ptr)
        self.freed.add((ptr, n))
        self.bytes_allocated -= n
        return 0

    def c_vd_memset(self, ptr, value, nbytes, stream):
        self.span('vd_memset', ptr, nbytes, 'destination')
        ctypes.memset(ptr, value, nbytes)
        return 0

    def c_vd_memcpy_h2d(self, dst, src, nbytes, stream):
        self.span('vd_memcpy_h2d', dst, nbytes, 'destination')
        if self.find(src) is not None:
            raise DryError('vd_memcpy_h2d: the source is device memory')
        ctypes.memmove(dst, src, nbytes)
        return 0

    def c_vd_memcpy_d2h(self, dst, src, nbytes, stream):
        self.span('vd_memcpy_d2h', src, nbytes, 'source')
        if self.find(dst) is not None:
            raise DryError('vd_memcpy_d2h: the destination is device memory')
        ctypes.memmove(dst, src, nbytes)
        return 0

    def c_vd_memcpy_d2d(self, dst, src, nbytes, stream):
        self.span('vd_memcpy_d2d', dst, nbytes, 'destination')
        self.span('vd_memcpy_d2d', src, nbytes, 'source')
        ctypes.memmove(dst, src, nbytes)
        return 0

    def c_vd_stream_synchronize(self, stream):
        return 0

    def c_vd_comm_unique_id(self, out):
        ctypes.memmove(out, bytes(range(128)), 128)
        return 0

    def c_vd_comm_init(self, rank, world, ident):
        if ctypes.string_at(ident, 128) != bytes(range(128)):
            return self.fail('dry: the rendezvous token is not the one vd_comm_unique_id produced')
        self.comm = (rank, world)
        return 0

    def c_vd_comm_available(self, version):
        if version:
            ctypes.cast(version, ctypes.POINTER(ctypes.c_int))[0] = 22203
        return 0

    def c_vd_comm_overlap_ms(self, lead):
        self.fail("vd_comm_overlap_ms: the last all-reduce was not split into an early and a late bucket")
        return -3

    def c_vd_comm_stats(self, b1, b2, ov, n):
        calls = getattr(self, 'comm_calls', 0)
        for ptr, ty, v in ((b1, ctypes.c_int64, 100 if calls else 0), (b2, ctypes.c_int64, 50 if calls else 0), (ov, ctypes.c_int, 1 if calls else 0),
                           (n, ctypes.c_int64, calls)):
            if ptr:
                ctypes.cast(ptr, ctypes.POINTER(ty))[0] = v
        return 0

    # ---- model level
    def _params_of(self, addr):
        st = self.ffi.structs.get('vd_model_params') or self.ffi.typedefs['vd_model_params']
        return {f[0]: _as_py(_struct_get(st, addr, f[0], None)) for f in st.fields}

    def c_vd_model_create(self, p, enc, dec, out):
        params = self._params_of(p)
        enc, dec = ctypes.string_at(enc).decode(), ctypes.string_at(dec).decode()
        try:
            spec = self.param_spec(enc, dec, params)
        except Exception as e:
            return self.fail('dry: unknown plug-in pair %s + %s (%s)' % (enc, dec, e))
        handle = ctypes.create_string_buffer(8)
        m = dict(handle=handle, params=params, enc=enc, dec=dec, spec=spec, training=1, lr=float(params['learningRate']),
                 W={n: np.zeros(s, np.float32) for n, s, _ in spec}, dW={n: np.zeros(s, np.float32) for n, s, _ in spec},
                 batch=None, steps=0, uploaded=0)
        self.models[ctypes.addressof(handle)] = m
        ctypes.c_void_p.from_address(out).value = ctypes.addressof(handle)
        return 0

    def c_vd_model_destroy(self, h):
        if h not in self.models:
            raise DryError('vd_model_destroy: not a live handle (destroyed twice?)')
        del self.models[h]

    def c_vd_model_num_tensors(self, h):
        return len(self.models[h]['spec'])

    def c_vd_model_flat_size(self, h):
        return sum((int(np.prod(s)) + 3) // 4 * 4 for _, s, _ in self.models[h]['spec'])

    def c_vd_model_tensor_info(self, h, i, name, off, rows, cols):
        spec = self.models[h]['spec']
        if not 0 <= i < len(spec):
            return self.fail('vd_model_tensor_info: index out of range')
        n, shape, _ = spec[i]
        ctypes.memmove(name, n.encode() + b'\0', len(n) + 1)
        o = sum((int(np.prod(s)) + 3) // 4 * 4 for _, s, _ in spec[:i])
        ctypes.c_int64.from_address(off).value = o
        ctypes.c_int64.from_address(rows).value = shape[0] if len(shape) == 2 else 1
        ctypes.c_int64.from_address(cols).value = shape[-1]
        return 0

    def c_vd_model_init_params(self, h, seed):
        rs = np.random.RandomState(seed % (2 ** 32))
        for n, w in self.models[h]['W'].items():
            w[...] = rs.uniform(-0.1, 0.1, w.shape)
        return 0

    def _tensor(self, h, name, which, n):
        m = self.models[h]
        name = ctypes.string_at(name).decode()
        t = (m['W'] if which == 0 else m['dW']).get(name)
        if t is None:
            return None, self.fail('no tensor named %s' % name)
        if t.size != n:
            return None, self.fail('tensor %s has %d elements, not %d' % (name, t.size, n))
        return t, 0

    def c_vd_model_set_tensor(self, h, name, host, n):
        t, rc = self._tensor(h, name, 0, n)
        if t is not None:
            ctypes.memmove(t.ctypes.data, host, 4 * n)
        return rc

    def c_vd_model_get_tensor(self, h, name, which, host, n):
        t, rc = self._tensor(h, name, which, n)
        if t is not None:
            ctypes.memmove(host, t.ctypes.data, 4 * n)
        return rc

    def c_vd_model_set_training(self, h, on):
        self.models[h]['training'] = on
        return 0

    def c_vd_model_upload_batch(self, h, b):
        st = self.ffi.typedefs['vd_batch']
        m = self.models[h]
        f = {x[0]: _as_py(_struct_get(st, b, x[0], None)) for x in st.fields}
        B, R = f['B'], m['params']['maxQuesCount']
        if B <= 0 or not f['ques_fwd']:
            return self.fail('vd_model_upload_batch: B <= 0 or no ques_fwd')
        got = {}
        for key, n, dt in (('ques_fwd', B * R * f['Tq'], np.int32), ('hist', B * R * f['Th'], np.int32),
                           ('answer_ind', B * R, np.int32), ('answer_in', B * R * f['Ta'], np.int32),
                           ('answer_out', B * R * f['Ta'], np.int32)):
            if f[key]:
                got[key] = np.frombuffer(ctypes.string_at(f[key], 4 * n), dt).copy()      # reads the host buffer NOW (consumed before return)
        if f['options']:
            O = m['params']['numOptions']
            got['options'] = np.frombuffer(ctypes.string_at(f['options'], 4 * B * R * O * f['To']), np.int32).copy()
        for k, v in got.items():
            if v.size and (v.min() < 0 or v.max() > m['params']['vocabSize'] + 1) and k != 'answer_ind':
                raise DryError('vd_model_upload_batch: %s holds values outside the vocabulary (min %d max %d): a dead or mistyped host buffer?'
                               % (k, v.min(), v.max()))
        if 'answer_ind' in got and (got['answer_ind'].min() < 1 or got['answer_ind'].max() > m['params']['numOptions']):
            raise DryError('vd_model_upload_batch: answer_ind must be 1-based in 1..numOptions (min %d max %d)'
                           % (got['answer_ind'].min(), got['answer_ind'].max()))
        m['batch'], m['dims'] = got, f
        m['uploaded'] += 1
        return 0

    def c_vd_model_forward_backward(self, h, only_forward):
        m = self.models[h]
        if m['batch'] is None:
            return self.fail('vd_model_forward_backward: no batch uploaded')
        m['steps'] += 1
        m['loss'] = 4.0 / m['steps']
        return 0

    def c_vd_model_loss(self, h, out):
        ctypes.c_float.from_address(out).value = self.models[h].get('loss', 0.0)
        return 0

    def c_vd_model_update(self, h, gscale):
        m = self.models[h]
        p = m['params']
        if m['lr'] > p['minLRate']:
            m['lr'] *= p['lrDecayRate']
        return 0

    def c_vd_model_learning_rate(self, h, lr, do_set):
        if do_set:
            self.models[h]['lr'] = ctypes.c_double.from_address(lr).value
        else:
            ctypes.c_double.from_address(lr).value = self.models[h]['lr']
        return 0

    def c_vd_model_retrieve(self, h):
        return 0 if self.models[h]['batch'] is not None else self.fail('no batch')

    def c_vd_model_ranks(self, h, use_gt, out):
        m = self.models[h]
        N, O = m['dims']['B'] * m['params']['maxQuesCount'], m['params']['numOptions']
        r = np.tile(np.arange(1, O + 1, dtype=np.int32), (N, 1))
        if use_gt:
            r = r[np.arange(N), m['batch']['answer_ind'] - 1].copy()
        ctypes.memmove(out, r.ctypes.data, r.nbytes)
        return 0

    def c_vd_model_encode(self, h):
        return 0

    def c_vd_model_decode_begin(self, h, rounds, n):
        self.models[h]['nlive'] = n
        return 0

    def c_vd_model_decode_step(self, h, toks, logp):
        m = self.models[h]
        V, n = m['params']['vocabSize'], m['nlive']
        t = np.frombuffer(ctypes.string_at(toks, 4 * n), np.int32)
        lp = np.full((n, V), -10.0, np.float32)
        for i in range(n):                                       # deterministic: token t is followed by t + 1, <END> = 2 closes
            lp[i, (int(t[i]) % V)] = -0.1 - 0.01 * i             # column c holds word id c + 1
            lp[i, (int(t[i]) + 3) % V] = -1.0
        ctypes.memmove(logp, lp.ctypes.data, lp.nbytes)
        return 0

    def c_vd_model_decode_select(self, h, src, n):
        return 0

    def c_vd_model_synchronize(self, h):
        return 0

    def c_vd_model_allreduce_grads(self, h):
        if not self.comm:
            return self.fail('vd_model_allreduce_grads: no communicator (vd_comm_init)')
        self.comm_calls = getattr(self, 'comm_calls', 0) + 1
        return 0


def _as_py(v):
    from .ffi import CInt64, CPointer
    if isinstance(v, CInt64):
        return v.val
    if isinstance(v, CPointer):
        return v.val
    return v
