"""Building blocks shared by the encoder plug-ins (each cites the reference lines it stands for).
No arithmetic here: every method is an ordered list of C-ABI launches."""
import numpy as np
import torch

from .. import ops
from ..nn import (SeqLSTM, Linear, dropout_forward, dropout_backward, lstm_stack_forward, lstm_stack_backward,
                  lstm2_bundle_forward, lstm2_bundle_backward)

P5 = 0.5            # the nngraph encoders hard-code Dropout(0.5) (mn-att:24,25,64,74,92,106)
S5 = 1.0 / (1.0 - P5)


def round_index(N, R, device, cache):
    """row n of a per-round tensor -> its image row n // R (model.lua:262-269 repeatTensor as a gather)"""
    idx = cache.get(('rep', N))
    if idx is None:
        idx = torch.from_numpy((np.arange(N) // R).astype(np.int32)).to(device)
        cache[('rep', N)] = idx
    return idx


class TextBranches(object):
    """history + question branches of the nngraph encoders: shared embedding -> Dropout(0.5) ->
    SeqLSTM(E,H):maskZero -> SeqLSTM(H,H):maskZero -> Select(1,-1)  (e.g. mn-att:21-45; two layers are
    hard-coded in those files).  Both stacks advance together as one skewed wavefront."""

    @staticmethod
    def declare(params, spec):
        E, H = params['embedSize'], params['rnnHiddenSize']
        spec.lstm('hist1', E, H); spec.lstm('hist2', H, H)
        spec.lstm('ques1', E, H); spec.lstm('ques2', H, H)

    def __init__(self, params, fp, ws, drop):
        E, H = params['embedSize'], params['rnnHiddenSize']
        self.E, self.H, self.fp, self.ws, self.drop = E, H, fp, ws, drop
        self.hist1, self.hist2 = SeqLSTM(fp, 'hist1', E, H, ws), SeqLSTM(fp, 'hist2', H, H, ws)
        self.ques1, self.ques2 = SeqLSTM(fp, 'ques1', E, H, ws), SeqLSTM(fp, 'ques2', H, H, ws)

    def _embed(self, name, tok):
        T, N = tok.shape
        m = self.drop.mask(name + '_emb', T * N * self.E, P5)
        x = self.ws.get(name + '.x', (T * N, self.E))
        ops.embed_gather(self.fp.w['embed'], tok, x, mask=m, scale=S5)
        return x, m

    def _stack_in(self, name, tok, l1, l2):
        """(bundle entry, sort meta) for one branch.  With length-sort metadata on the token tensor
        (Model.prepare_inputs) the stack runs on length-sorted rows and skips pad (t, row) pairs."""
        T, N = tok.shape
        x, m = self._embed(name, tok)
        srt = getattr(tok, 'vd_sort', None)
        if srt is None:
            return (l1, l2, x, T, N, tok), None, m
        xs = ops.embed_gather(x, srt.fwd_idx, self.ws.get(name + '.xs', (T * N, self.E)))   # permute rows per step
        return (l1, l2, xs, T, N, srt.tok_sorted, srt.nact, srt.nact_dev, srt.rows), srt, m

    def forward(self, ques, hist):
        N, H = ques.shape[1], self.H
        eh, self.s_h, self.m_h = self._stack_in('h', hist, self.hist1, self.hist2)
        eq, self.s_q, self.m_q = self._stack_in('q', ques, self.ques1, self.ques2)
        hT, qT = hist.shape[0], ques.shape[0]
        hh, qh = lstm2_bundle_forward([eh, eq])
        h3, q3 = hh[hT - 1], qh[qT - 1]
        if self.s_h is not None:
            h3 = ops.embed_gather(h3, self.s_h.inv, self.ws.get('h.last', (N, H)))          # back to batch order
        if self.s_q is not None:
            q3 = ops.embed_gather(q3, self.s_q.inv, self.ws.get('q.last', (N, H)))
        return q3, h3

    def backward(self, ques, hist, dq3, dh3):
        N, H, E = ques.shape[1], self.H, self.E
        if self.s_h is not None:
            dh3 = ops.embed_gather(dh3, self.s_h.perm, self.ws.get('h.dlast', (N, H)))
        if self.s_q is not None:
            dq3 = ops.embed_gather(dq3, self.s_q.perm, self.ws.get('q.dlast', (N, H)))
        dhx, dqx = lstm2_bundle_backward([(self.hist1, self.hist2, dh3), (self.ques1, self.ques2, dq3)])
        if self.s_h is not None:
            dhx = ops.embed_gather(dhx, self.s_h.inv_idx, self.ws.get('h.dxo', (hist.numel(), E)))
        if self.s_q is not None:
            dqx = ops.embed_gather(dqx, self.s_q.inv_idx, self.ws.get('q.dxo', (ques.numel(), E)))
        ops.embed_scatter_acc(self.fp.g['embed'], hist, dhx, mask=self.m_h, scale=S5)
        ops.embed_scatter_acc(self.fp.g['embed'], ques, dqx, mask=self.m_q, scale=S5)


class MemoryBlock(object):
    """nn.MM(false,true) -> MaskSoftMax -> nn.MM -> Tanh(Linear(Dropout)) -> Tanh(Linear(hAttTr + query))
    (mn-att:48-65; mn-ques-hist.lua:43-58; mn-ques-im-hist.lua:50-65)."""

    @staticmethod
    def declare(params, spec):
        H = params['rnnHiddenSize']
        spec.linear('mn1', H, H); spec.linear('mn2', H, H)

    def __init__(self, params, fp, ws, drop):
        self.H, self.R, self.ws, self.drop = params['rnnHiddenSize'], params['maxQuesCount'], ws, drop
        self.mn1, self.mn2 = Linear(fp, 'mn1', self.H, self.H, ws), Linear(fp, 'mn2', self.H, self.H, ws)

    def forward(self, query, h3, mask):
        ws, H, R = self.ws, self.H, self.R
        N = query.shape[0]
        B = N // R
        self.query, self.h3, self.N, self.B = query, h3, N, B
        self.prob = ws.get('mn.prob', (N, R))
        hatt = ws.get('mn.hatt', (N, H))
        ops.mn_attention_forward(query, h3, mask, self.prob, hatt, B, R, H)
        self.m_hatt = self.drop.mask('hatt', N * H, P5)
        hatt_d = dropout_forward(ws, 'mn.hatt_d', hatt, self.m_hatt, S5)
        hattTr = self.mn1.forward(hatt_d, N, tanh=True)
        s2 = ops.axpby(hattTr, query, ws.get('mn.s2', (N, H)), 1.0, 1.0)         # CAddTable
        return self.mn2.forward(s2, N, tanh=True)

    def backward(self, dqh2):
        """returns (d query, d h3)"""
        ws, H, R, N, B = self.ws, self.H, self.R, self.N, self.B
        ds2 = self.mn2.backward(dqh2)
        dhatt_d = self.mn1.backward(ds2)
        dhatt = dropout_backward(ws, 'mn.dhatt', dhatt_d, self.m_hatt, S5)
        dq_att = ws.get('mn.dq', (N, H))
        dh3 = ws.get('mn.dh', (N, H))
        ops.mn_attention_backward(self.query, self.h3, self.prob, dhatt, dq_att, dh3, B, R, H)
        dquery = ops.axpby(dq_att, ds2, ws.get('mn.dquery', (N, H)), 1.0, 1.0)
        return dquery, dh3


class SANBlock(object):
    """Stacked attention over the S x S image regions + output layer (mn-att:68-106; lf-att-ques-im-hist.lua:45-86),
    `numAttentionLayers` hops (opts.lua:26, default 1): every hop has its OWN img_common / ques_common / att Linears
    and its own Dropout on the joint embedding, all hops attend over the same img_tr (one Dropout mask), and
    u_i = sum_s p_i[s] img_tr[s] + u_{i-1} (mn-att:82-104).  `pre` = tanh(Linear(img)) is computed once per IMAGE; the
    per-round Dropout masks are applied by the GEMM loaders, the 10x repeat (model.lua:262-265) never materialises.
    Parameter names: hop 1 = img_common / ques_common / att, hop i > 1 = img_common<i> / ques_common<i> / att<i>."""

    @staticmethod
    def hop_names(params):
        L = int(params.get('numAttentionLayers', 1) or 1)
        if params.get('encoder') == 'lf-att-ques-im-hist':          # lf-att-ques-im-hist.lua:49 hard-codes one hop
            L = 1
        return [('img_common' + s, 'ques_common' + s, 'att' + s) for s in [''] + [str(i) for i in range(2, L + 1)]]

    @staticmethod
    def declare(params, spec):
        H, C, K = params['rnnHiddenSize'], params['imgFeatureSize'], params.get('commonEmbeddingSize', 512)
        spec.linear('img_proj', C, H)
        for ic, qc, at in SANBlock.hop_names(params):
            spec.linear(ic, H, K)
            spec.linear(qc, H, K)
            spec.linear(at, K, 1)
        spec.linear('out', H, H)

    def __init__(self, params, fp, ws, drop, streams):
        self.fp, self.ws, self.drop, self.streams = fp, ws, drop, streams
        self.H, self.C = params['rnnHiddenSize'], params['imgFeatureSize']
        self.K = params.get('commonEmbeddingSize', 512)
        self.S2, self.R = params['imgSpatialSize'] ** 2, params['maxQuesCount']
        self.hops = SANBlock.hop_names(params)
        self.img_proj = Linear(fp, 'img_proj', self.C, self.H, ws)
        self.ques_common = [Linear(fp, qc, self.H, self.K, ws) for _, qc, _ in self.hops]
        self.out = Linear(fp, 'out', self.H, self.H, ws)

    @staticmethod
    def _sfx(i):
        return '' if i == 0 else str(i + 1)

    def prefetch(self, img, N):
        """per-image projection + this step's dropout masks: independent of the text branches, so it is
        enqueued on its own stream before them"""
        H, K, S2 = self.H, self.K, self.S2
        B = N // self.R
        with self.streams.fork('img'):
            self.pre = self.img_proj.forward(img, B * S2, tanh=True)             # mn-att:74-78 (pre-dropout)
            self.m1 = self.drop.mask('img_tr', N * S2 * H, P5)
            self.m2 = [self.drop.mask('iqc' + self._sfx(i), N * S2 * K, P5) for i in range(len(self.hops))]

    def forward(self, u0):
        fp, ws, H, K, S2, R = self.fp, self.ws, self.H, self.K, self.S2, self.R
        N = u0.shape[0]
        self.N = N
        self.streams.join('img')
        sc = S5 if self.m1 is not None else 1.0
        self.sc = sc
        self.u_in, self.iqc, self.patt = [], [], []
        u = u0
        for i, (ic, _, at) in enumerate(self.hops):
            sfx = self._sfx(i)
            qc = self.ques_common[i].forward(u, N)                               # mn-att:88
            iqc = ws.get('att.iqc' + sfx, (N * S2, K))
            ops.img_common_forward(self.pre, self.m1, fp.w[ic + '.W'], fp.w[ic + '.b'], qc, self.m2[i], iqc, N, R, S2, H,
                                   K, sc)                                        # mn-att:83-92
            patt = ws.get('att.p' + sfx, (N, S2))
            u1 = ws.get('att.u1' + sfx, (N, H))
            ops.img_att_forward(iqc, fp.w[at + '.W'], fp.w[at + '.b'], self.pre, self.m1, u, patt, u1, N, R, S2, H, K,
                                sc)                                              # mn-att:93-102
            self.u_in.append(u); self.iqc.append(iqc); self.patt.append(patt)
            u = u1
        self.m_u = self.drop.mask('u', N * H, P5)
        u_d = dropout_forward(ws, 'att.u1_d', u, self.m_u, S5)
        return self.out.forward(u_d, N, tanh=True)                               # mn-att:106

    def backward(self, grad_output):
        """returns d u0"""
        fp, ws, H, K, S2, R, N, sc = self.fp, self.ws, self.H, self.K, self.S2, self.R, self.N, self.sc
        B = N // R
        G = fp.g
        du1d = self.out.backward(grad_output)
        du = dropout_backward(ws, 'att.du1', du1d, self.m_u, S5)                 # = d att of the last hop + its residual
        dpre = ws.get('att.dpre', (B * S2, H))
        ops.zero(dpre)
        for i in range(len(self.hops) - 1, -1, -1):
            ic, _, at = self.hops[i]
            sfx = self._sfx(i)
            dqc = ws.get('att.dqc' + sfx, (N, K))
            ops.img_att_backward(self.iqc[i], fp.w[at + '.W'], self.pre, self.m1, self.m2[i], self.patt[i], du,
                                 G[at + '.W'], G[at + '.b'], dqc, ws.get('att.dscore', (N, S2)), N, R, S2, H, K, sc)
            dz = self.iqc[i]                                                     # iqc now holds dz
            ops.colsum_acc(dz, G[ic + '.b'], M=N * S2, N=K)
            ops.img_common_wgrad(dz, self.pre, self.m1, G[ic + '.W'], N, R, S2, H, K, sc)
            ops.img_tr_backward(dz, fp.w[ic + '.W'], self.patt[i], du, self.m1, dpre, N, R, S2, H, K, sc)   # += into dpre
            du_q = self.ques_common[i].backward(dqc)
            du = ops.axpby(du_q, du, ws.get('att.du0' + sfx, (N, H)), 1.0, 1.0)   # residual CAddTable (mn-att:102)
        self.img_proj.backward(dpre, need_dx=False)                              # tanh' + dW, db of mn-att:77
        return du


class CatLinear(object):
    """Tanh(Linear(JoinTable(parts))) with optional Dropout(p) on the joined vector:
    lf-*.lua fuse layer, mn-ques-im-hist.lua:47-48 (qi), lf-att-ques-im-hist.lua:43 (qh)."""

    def __init__(self, fp, name, dims, H, ws):
        self.dims, self.D, self.H, self.ws, self.name = list(dims), sum(dims), H, ws, name
        self.lin = Linear(fp, name, self.D, H, ws)

    def forward(self, parts, N, mask=None, scale=1.0):
        ws, D = self.ws, self.D
        cat = ws.get(self.name + '.cat', (N, D))
        off = 0
        for x, d in zip(parts, self.dims):
            ops.copy_2d(cat, D, x, d, N, d, dst_off=off)
            off += d
        self.N, self.mask, self.scale = N, mask, scale
        catd = dropout_forward(ws, self.name + '.in', cat, mask, scale)
        return self.lin.forward(catd, N, tanh=True)

    def backward(self, dy, need=None):
        """returns the list of per-part gradients ([N x d_i] contiguous; None where not needed)"""
        ws, D, N = self.ws, self.D, self.N
        dcatd = self.lin.backward(dy)
        dcat = dropout_backward(ws, self.name + '.din', dcatd, self.mask, self.scale)
        outs, off = [], 0
        need = need or [True] * len(self.dims)
        for i, d in enumerate(self.dims):
            if need[i]:
                g = ws.get('%s.dpart%d' % (self.name, i), (N, d))
                ops.copy_2d(g, d, dcat, D, N, d, src_off=off)
                outs.append(g)
            else:
                outs.append(None)
            off += d
        return outs
