"""Host-side building blocks the encoder/decoder plug-ins are composed from.

These play the role Torch's nn / rnn modules play for the reference's encoders/*.lua and
decoders/*.lua: they own no arithmetic -- every forward/backward is a call into the C ABI
(visdial_amd.ops) -- only buffers, shapes and the order of kernel launches.
"""
import zlib

import torch

from . import ops

F32, I32, U8 = torch.float32, torch.int32, torch.uint8


class Workspace(object):
    """Persistent named device buffers: allocated once at the largest size seen, handed out as
    contiguous prefix views (time-major activations of shorter batches are prefixes)."""

    def __init__(self, device):
        self.device = device
        self.bufs = {}

    def get(self, name, shape, dtype=F32):
        n = 1
        for s in shape:
            n *= int(s)
        buf = self.bufs.get(name)
        if buf is None or buf.numel() < n or buf.dtype != dtype:
            buf = torch.empty(max(n, 1), dtype=dtype, device=self.device)
            self.bufs[name] = buf
        return buf[:n].view(*[int(s) for s in shape])

    def nbytes(self):
        return sum(b.numel() * b.element_size() for b in self.bufs.values())


class StreamPool(object):
    """Fork/join helper over HIP streams (torch only provides the stream/event handles).

    The step has three independent launch chains -- history LSTMs, question LSTMs (both
    latency-bound, ~120 dependent ~30 us launches) and the option LSTM (throughput-bound, all CUs)
    -- so they are enqueued on separate streams and the hardware dispatcher interleaves the small
    chains under the big MFMA kernels.  Side streams get high priority so their short kernels are
    dispatched as soon as a CU has room."""

    def __init__(self, device, enabled=True):
        self.enabled = enabled
        self.device = device
        self.side = {}

    def get(self, name):
        s = self.side.get(name)
        if s is None:
            s = torch.cuda.Stream(device=self.device, priority=-1)
            self.side[name] = s
        return s

    def fork(self, name, after=None):
        """Return a context manager that makes `name` the current stream, ordered after everything
        already enqueued on the current stream (or, with `after`, only after that recorded event).
        Disabled pool: a no-op context."""
        if not self.enabled:
            return _NullCtx()
        s = self.get(name)
        if after is not None:
            s.wait_event(after)
        else:
            s.wait_stream(torch.cuda.current_stream())
        return torch.cuda.stream(s)

    def join(self, name):
        """Make the current stream wait for everything enqueued on `name`."""
        if self.enabled:
            torch.cuda.current_stream().wait_stream(self.get(name))


class _NullCtx(object):
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


class DropoutState(object):
    """nn.Dropout noise.  training + no external masks: keep-masks come from the on-device
    counter-based generator (fresh seed per call site per step); `external` (dict name -> uint8
    device tensor) pins them for parity tests; evaluate(): no masks (identity)."""

    def __init__(self, ws, seed=1234):
        self.ws = ws
        self.seed = seed
        self.step = 0
        self.training = True
        self.external = None

    def mask(self, name, numel, p):
        if not self.training or p <= 0.0:
            return None
        if self.external is not None:
            m = self.external[name]
            assert m.numel() == numel and m.dtype == U8, name
            return m
        m = self.ws.get('dropmask.' + name, (numel,), U8)
        ops.dropout_mask(m, (self.seed * 1000003 + self.step) * 64 + (zlib.crc32(name.encode()) & 63), p)
        return m

    def next_step(self):
        self.step += 1


class SeqLSTM(object):
    """nn.SeqLSTM(D, H) [:maskZero()] -- forward/backward over a whole time-major sequence.
    Saved state (gates, h, c) lives in the workspace under `key`."""

    def __init__(self, fp, name, D, H, ws, key=None):
        self.D, self.H = D, H
        self.W, self.b = fp.w[name + '.W'], fp.w[name + '.b']
        self.dW, self.db = fp.g[name + '.W'], fp.g[name + '.b']
        self.Wx, self.Wh = self.W[:D], self.W[D:]
        self.dWx, self.dWh = self.dW[:D], self.dW[D:]
        self.ws = ws
        self.key = key or name

    def forward(self, x, T, N, tok_mask=None, h0=None, c0=None):
        """x: [T*N, D] dense input rows (time-major).  Returns h [T, N, H]."""
        H, k = self.H, self.key
        self.T, self.N, self.x, self.tok_mask, self.h0, self.c0 = T, N, x, tok_mask, h0, c0
        self.gates = self.ws.get(k + '.gates', (T, N, 4 * H))
        self.h = self.ws.get(k + '.h', (T, N, H))
        self.c = self.ws.get(k + '.c', (T, N, H))
        g2 = self.gates.view(T * N, 4 * H)
        # hoisted input projection, written straight into the gates buffer (the step kernel reads
        # and overwrites each element from the same thread, so the recurrence runs in place)
        ops.gemm_nn(x, self.Wx, g2, bias=self.b, M=T * N, N=4 * H, K=self.D)
        ops.lstm_forward(g2, self.Wh, self.gates, self.h, self.c, T, N, H, N * 4 * H, 4 * H, tok_mask=tok_mask,
                         h0=h0, c0=c0)
        return self.h

    def backward(self, dh_seq=None, dh_last=None, dc_last=None, need_dx=True, need_dh0=False):
        """Returns (dx [T*N, D] or None, dh0 or None, dc0).  Accumulates dW, db."""
        T, N, H, k = self.T, self.N, self.H, self.key
        dc = self.ws.get(k + '.dc', (N, H))
        dh0 = self.ws.get(k + '.dh0', (N, H)) if need_dh0 else None
        ops.lstm_backward(self.Wh, self.gates, self.c, dc, T, N, H, c0=self.c0, dh_seq=dh_seq, dh_last=dh_last,
                          dc_last=dc_last, dh0=dh0)
        da = self.gates.view(T * N, 4 * H)
        h2 = self.h.view(T * N, H)
        if T > 1:
            ops.gemm_tn_acc(h2, da[N:], self.dWh, M=H, N=4 * H, K=(T - 1) * N)
        if self.h0 is not None:
            ops.gemm_tn_acc(self.h0, da, self.dWh, M=H, N=4 * H, K=N)
        ops.gemm_tn_acc(self.x, da, self.dWx, M=self.D, N=4 * H, K=T * N)
        ops.colsum_acc(da, self.db, M=T * N, N=4 * H)
        dx = None
        if need_dx:
            dx = self.ws.get(k + '.dx', (T * N, self.D))
            ops.gemm_nt(da, self.Wx, dx, M=T * N, N=self.D, K=4 * H)
        return dx, dh0, dc


class Linear(object):
    """nn.Linear (+ optional fused nn.Tanh)."""

    def __init__(self, fp, name, n_in, n_out, ws, key=None):
        self.n_in, self.n_out = n_in, n_out
        self.W, self.b = fp.w[name + '.W'], fp.w[name + '.b']
        self.dW, self.db = fp.g[name + '.W'], fp.g[name + '.b']
        self.ws = ws
        self.key = key or name

    def forward(self, x, M, tanh=False):
        self.x, self.M, self.tanh = x, M, tanh
        self.y = self.ws.get(self.key + '.y', (M, self.n_out))
        ops.gemm_nt(x, self.W, self.y, bias=self.b, act=1 if tanh else 0, M=M, N=self.n_out, K=self.n_in)
        return self.y

    def backward(self, dy, need_dx=True):
        M = self.M
        if self.tanh:
            dpre = self.ws.get(self.key + '.dpre', (M, self.n_out))
            ops.tanh_backward(dy, self.y, dpre)
        else:
            dpre = dy
        ops.gemm_tn_acc(dpre, self.x, self.dW, M=self.n_out, N=self.n_in, K=M)
        ops.colsum_acc(dpre, self.db, M=M, N=self.n_out)
        if not need_dx:
            return None
        dx = self.ws.get(self.key + '.dx', (M, self.n_in))
        ops.gemm_nn(dpre, self.W, dx, M=M, N=self.n_in, K=self.n_out)
        return dx


def dropout_forward(ws, key, x, mask, scale):
    """y = dropout(x) (identity object when mask is None)."""
    if mask is None:
        return x
    y = ws.get(key, tuple(x.shape))
    ops.dropout_apply(x, mask, y, scale)
    return y


def dropout_backward(ws, key, dy, mask, scale):
    if mask is None:
        return dy
    dx = ws.get(key, tuple(dy.shape))
    ops.dropout_apply(dy, mask, dx, scale)
    return dx
