"""Host-side building blocks the encoder/decoder plug-ins are composed from.

These play the role Torch's nn / rnn modules play for the reference's encoders/*.lua and
decoders/*.lua: they own no arithmetic -- every forward/backward is a call into the C ABI
(visdial_amd.ops) -- only buffers, shapes and the order of kernel launches.
"""
import zlib

import os

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
        self._sites = {}

    def mask(self, name, numel, p):
        if not self.training or p <= 0.0:
            return None
        if self.external is not None:
            m = self.external[name]
            assert m.numel() == numel and m.dtype == U8, name
            return m
        m = self.ws.get('dropmask.' + name, (numel,), U8)
        # one independent counter stream per (seed, step, call site): the full 32-bit crc of the site name goes into
        # the 64-bit generator seed (6 bits of it once made 'h_emb' and 'hatt' share a stream)
        site = zlib.crc32(name.encode()) & 0xffffffff
        clash = self._sites.setdefault(site, name)
        assert clash == name, "dropout sites %r and %r hash to the same stream" % (clash, name)
        ops.dropout_mask(m, (((self.seed * 1000003 + self.step) & 0xffffffff) << 32) | site, p)
        return m

    def next_step(self):
        self.step += 1


class SeqLSTM(object):
    """nn.SeqLSTM(D, H) [:maskZero()] -- forward/backward over a whole time-major sequence.

    The input may arrive as several column blocks (`part_dims`): x*Wx is then the sum of the
    per-block products against the matching row blocks of Wx, which is how nn.JoinTable(-1) in front
    of an LSTM (encoders/hre-ques-im-hist.lua:66-68,85-92) is folded away without a concat copy.
    State hand-off fields carry the reference's names (rnn SeqLSTM, used by decoders/gen.lua:30-60):
    userPrevOutput / userPrevCell (h0, c0), userNextGradCell (dc at the last step), gradPrevOutput
    (extra dh at the last step); after backward userGradPrevOutput / userGradPrevCell (dh0, dc0).
    Saved state (gates, h, c) lives in the workspace under `key`: `.output` = h, `.cell` = c."""

    def __init__(self, fp, name, D, H, ws, key=None, part_dims=None):
        self.D, self.H = D, H
        self.part_dims = list(part_dims) if part_dims else [D]
        assert sum(self.part_dims) == D
        self.W, self.b = fp.w[name + '.W'], fp.w[name + '.b']
        self.dW, self.db = fp.g[name + '.W'], fp.g[name + '.b']
        self.Wx, self.Wh = self.W[:D], self.W[D:]
        self.dWx, self.dWh = self.dW[:D], self.dW[D:]
        self.ws = ws
        self.key = key or name
        self.userPrevOutput = self.userPrevCell = None
        self.userNextGradCell = self.gradPrevOutput = None
        self.userGradPrevOutput = self.userGradPrevCell = None

    def _wx_blocks(self, W):
        out, o = [], 0
        for d in self.part_dims:
            out.append(W[o:o + d])
            o += d
        return out

    def forward(self, x, T, N, tok_mask=None):
        """x: [T*N, D] dense input rows (time-major), or a list of column blocks.  Returns h [T, N, H]."""
        H, k = self.H, self.key
        xs = list(x) if isinstance(x, (list, tuple)) else [x]
        assert len(xs) == len(self.part_dims)
        h0, c0 = self.userPrevOutput, self.userPrevCell
        self.userPrevOutput = self.userPrevCell = None          # consumed once (rnn semantics)
        if h0 is not None and c0 is None:
            c0 = self.ws.get(k + '.c0zero', (N, H))
            ops.zero(c0)
        if c0 is not None and h0 is None:
            h0 = self.ws.get(k + '.h0zero', (N, H))
            ops.zero(h0)
        self.T, self.N, self.xs, self.tok_mask, self.h0, self.c0 = T, N, xs, tok_mask, h0, c0
        self.rows = None
        self.gates = self.ws.get(k + '.gates', (T, N, 4 * H))
        self.h = self.ws.get(k + '.h', (T, N, H))
        self.c = self.ws.get(k + '.c', (T, N, H))
        self.output, self.cell = self.h, self.c
        g2 = self.gates.view(T * N, 4 * H)
        # hoisted input projection, written straight into the gates buffer (the step kernel reads
        # and overwrites each element from the same thread, so the recurrence runs in place)
        for i, (xi, wi, d) in enumerate(zip(xs, self._wx_blocks(self.Wx), self.part_dims)):
            ops.gemm_nn(xi, wi, g2, bias=self.b if i == 0 else None, accumulate=(i > 0), M=T * N, N=4 * H, K=d)
        ops.lstm_forward(g2, self.Wh, self.gates, self.h, self.c, T, N, H, N * 4 * H, 4 * H, tok_mask=tok_mask,
                         h0=h0, c0=c0)
        return self.h

    def backward(self, dh_seq=None, dh_last=None, need_dx=True):
        """Returns the list of dx blocks ([T*N, d_i]; None where not requested).  Accumulates dW, db and
        sets userGradPrevOutput / userGradPrevCell."""
        T, N, H, k = self.T, self.N, self.H, self.key
        if self.gradPrevOutput is not None:
            if dh_last is None:
                dh_last = self.gradPrevOutput
            else:
                dh_last = ops.axpby(dh_last, self.gradPrevOutput, self.ws.get(k + '.dhl', (N, H)), 1.0, 1.0)
        dc_last = self.userNextGradCell
        self.gradPrevOutput = self.userNextGradCell = None
        dc = self.ws.get(k + '.dc', (N, H))
        dh0 = self.ws.get(k + '.dh0', (N, H)) if self.h0 is not None else None
        ops.lstm_backward(self.Wh, self.gates, self.c, dc, T, N, H, c0=self.c0, dh_seq=dh_seq, dh_last=dh_last,
                          dc_last=dc_last, dh0=dh0)
        self.userGradPrevOutput, self.userGradPrevCell = dh0, (dc if self.h0 is not None else None)
        return self.param_grads(need_dx)

    def param_grads(self, need_dx=True):
        """Weight/bias gradients and input gradients from da (held in the gates buffer after BPTT)."""
        T, N, H, k = self.T, self.N, self.H, self.key
        da = self.gates.view(T * N, 4 * H)
        h2 = self.h.view(T * N, H)
        rows = getattr(self, 'rows', None)          # (act, act1, prev1): contract the non-pad (t, row) pairs only
        # ... except where the shape fits the k-major LDS-DMA contraction (M, N multiples of 128): there the DENSE
        # product over all T*N rows is faster than the index-list kernel (pad pairs hold da = 0) -- same rule as
        # csrc/rt_core.h SeqLSTM::param_grads
        fits = lambda m, kk: m % 128 == 0 and (4 * H) % 128 == 0 and kk >= 1024
        if rows is not None and not (T > 1 and fits(H, (T - 1) * N)):
            act, act1, prev1 = rows
            if act1.numel():
                ops.gemm_tn_rows_acc(h2, prev1, da, act1, self.dWh, M=H, N=4 * H)
        elif T > 1:
            ops.gemm_tn_acc(h2, da[N:], self.dWh, M=H, N=4 * H, K=(T - 1) * N)
        if self.h0 is not None:
            ops.gemm_tn_acc(self.h0, da, self.dWh, M=H, N=4 * H, K=N)
        ops.colsum_acc(da, self.db, M=T * N, N=4 * H)
        need = need_dx if isinstance(need_dx, (list, tuple)) else [need_dx] * len(self.xs)
        dxs = []
        for i, (xi, wi, dwi, d) in enumerate(zip(self.xs, self._wx_blocks(self.Wx), self._wx_blocks(self.dWx),
                                                self.part_dims)):
            if rows is not None and not fits(d, T * N):
                ops.gemm_tn_rows_acc(xi, rows[0], da, rows[0], dwi, M=d, N=4 * H)
            else:
                ops.gemm_tn_acc(xi, da, dwi, M=d, N=4 * H, K=T * N)
            if need[i]:
                dx = self.ws.get('%s.dx%d' % (k, i), (T * N, d))
                ops.gemm_nt(da, wi, dx, M=T * N, N=d, K=4 * H)
                dxs.append(dx)
            else:
                dxs.append(None)
        return dxs


class SeqSort(object):
    """Length-sorted view of a right-aligned [T x N] token matrix (host-side metadata computed where the
    batch is still in host memory).  Rows are ordered by decreasing length, so at step t the non-pad rows
    are the prefix [0, nact[t]) and the recurrence kernels skip every pad (t, row) pair.
      tok_sorted  device int32 [T x N]     token matrix with permuted columns
      fwd_idx     device int32 [T*N]       row of the ORIGINAL [T*N x .] tensor feeding sorted row i
      inv_idx     device int32 [T*N]       row of the SORTED tensor feeding original row i
      perm / inv  device int32 [N]         the same permutations for [N x .] tensors
      nact        host   int32 [T], nact_dev its device copy"""

    def __init__(self, tok_host, device):
        import numpy as np
        T, N = tok_host.shape
        length = (tok_host != 0).sum(0)
        perm = np.argsort(-length, kind='stable').astype(np.int32)
        inv = np.empty(N, np.int32)
        inv[perm] = np.arange(N, dtype=np.int32)
        self.nact = np.ascontiguousarray((length[None, :] >= (T - np.arange(T))[:, None]).sum(1).astype(np.int32))
        base = (np.arange(T, dtype=np.int32) * N)[:, None]
        dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(device)
        self.tok_sorted = dev(tok_host[:, perm].astype(np.int32))
        self.fwd_idx = dev((base + perm[None, :]).reshape(-1))
        self.inv_idx = dev((base + inv[None, :]).reshape(-1))
        self.perm, self.inv, self.nact_dev = dev(perm), dev(inv), dev(self.nact)
        self.T, self.N = T, N
        # the non-pad (t, row) pairs as linear row indices of the sorted [T*N x .] tensors, in step order: the
        # weight-gradient contractions run over these only (vd_gemm_tn_rows_acc).  act1 / prev1 = the pairs with
        # t >= 1 and their previous-step rows (dWh = sum_t h_{t-1}^T da_t).
        act = np.concatenate([t * N + np.arange(self.nact[t], dtype=np.int32) for t in range(T)]).astype(np.int32) \
            if T else np.zeros(0, np.int32)
        n0 = int(self.nact[0]) if T else 0
        self.rows = None
        if act.size:
            self.rows = (dev(act), dev(act[n0:]), dev(act[n0:] - N))


def lstm2_bundle_forward(bundle):
    """bundle: list of (l1, l2, x, T, N, tok_mask[, nact, nact_dev]) -- two-layer maskZero stacks advanced
    together as a skewed wavefront (vd_lstm2_forward): one grouped launch per tick for ALL stacks.  With
    nact (rows sorted by length) pad rows are skipped and zero-filled; an optional 9th item `rows` = (act, act1, prev1)
    row lists (SeqSort.rows) lets the weight-gradient contractions skip the pad pairs too.  Returns the top layers' h tensors.
    Fills the same saved-state fields SeqLSTM.forward does."""
    H = bundle[0][0].H
    descs = []
    for item in bundle:
        l1, l2, x, T, N, tok = item[:6]
        nact, nact_dev = (item[6], item[7]) if len(item) > 6 else (None, None)
        l1.rows = l2.rows = item[8] if len(item) > 8 else None
        assert l1.H == H and l2.H == H and l2.D == H and len(l1.part_dims) == 1
        for l, xs in ((l1, [x]), (l2, None)):
            l.T, l.N, l.tok_mask, l.h0, l.c0 = T, N, tok, None, None
            l.gates = l.ws.get(l.key + '.gates', (T, N, 4 * H))
            l.h = l.ws.get(l.key + '.h', (T, N, H))
            l.c = l.ws.get(l.key + '.c', (T, N, H))
            l.output, l.cell = l.h, l.c
            l.userPrevOutput = l.userPrevCell = None
        l1.xs = [x]
        l2.xs = [l1.h.view(T * N, H)]
        ops.gemm_nn(x, l1.Wx, l1.gates.view(T * N, 4 * H), bias=l1.b, M=T * N, N=4 * H, K=l1.D)
        l1.nact = l2.nact = nact
        if nact is not None:
            # skipped (t, row) pairs must read as zeros: previous state of rows that become active later,
            # and da = 0 in the weight-gradient contractions
            ops.zero_inactive_rows(l1.gates, nact_dev, T, N, 4 * H)
            for b in (l1.h, l1.c, l2.h, l2.c, l2.gates):
                ops.zero(b)
        descs.append(dict(T=T, N=N, tok_mask=tok, Wh1=l1.Wh, Wx2=l2.Wx, b2=l2.b, Wh2=l2.Wh, gates1=l1.gates, h1=l1.h,
                          c1=l1.c, gates2=l2.gates, h2=l2.h, c2=l2.c, nact=nact))
    ops.lstm2_forward(descs, H)
    return [b[1].h for b in bundle]


def lstm2_bundle_backward(bundle):
    """bundle: list of (l1, l2, dh_last_top).  Runs the fused two-layer BPTT for all stacks, then the
    weight-gradient contractions per layer.  Returns the bottom layers' dx ([T*N, D])."""
    H = bundle[0][0].H
    descs = []
    for l1, l2, dlast in bundle:
        T, N = l1.T, l1.N
        ws = l1.ws
        descs.append(dict(T=T, N=N, Wh1=l1.Wh, Wx2=l2.Wx, Wh2=l2.Wh, gates1=l1.gates, c1=l1.c, gates2=l2.gates,
                          c2=l2.c, dh_last2=dlast, dh1_seq=ws.get(l1.key + '.dhseq', (T, N, H)),
                          dc1=ws.get(l1.key + '.dc', (N, H)), dc2=ws.get(l2.key + '.dc', (N, H)),
                          nact=getattr(l1, 'nact', None)))
    ops.lstm2_backward(descs, H)
    out = []
    for l1, l2, _ in bundle:
        l2.param_grads(need_dx=False)
        out.append(l1.param_grads(need_dx=True)[0])
    return out


def lstm_stack_forward(layers, x, T, N, tok_mask):
    """numLayers x SeqLSTM(maskZero) as in encoders/lf-ques.lua:18-24.  Returns the top layer's h."""
    h = layers[0].forward(x, T, N, tok_mask)
    for l in layers[1:]:
        h = l.forward(h.view(T * N, l.D), T, N, tok_mask)
    return h


def lstm_stack_backward(layers, dh_last_top=None, dh_seq_top=None, need_dx=True):
    """Backward through the stack; returns the bottom layer's dx blocks."""
    dseq = dh_seq_top
    for i in range(len(layers) - 1, -1, -1):
        l = layers[i]
        top = (i == len(layers) - 1)
        dxs = l.backward(dh_seq=dseq, dh_last=dh_last_top if top else None, need_dx=need_dx if i == 0 else True)
        if i > 0:
            dseq = dxs[0].view(l.T, l.N, layers[i - 1].H)
    return dxs


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
