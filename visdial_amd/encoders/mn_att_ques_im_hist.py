"""Memory-network + image-attention encoder: counterpart of encoders/mn-att-ques-im-hist.lua:5-115.

Inputs (order of the reference's input table, model.lua:255-294):
  ques [Tq x N] int32 time-major, img [B*S2 x C] fp32 (one map per IMAGE -- the reference's 10x
  repeatTensor copy, model.lua:262-265, is folded into the attention kernels' loaders), hist [Th x N],
  mask [N x R] uint8 (1 = hidden).  Output encOut [N x H].
Like the reference this encoder hard-codes two LSTM layers per branch and dropout 0.5 (mn-att:24-41).
"""
from .. import ops
import os

from ..nn import (SeqLSTM, Linear, StreamPool, dropout_forward, dropout_backward, lstm_stack_backward,
                  lstm2_bundle_forward, lstm2_bundle_backward)

P_DROP = 0.5
SCALE = 1.0 / (1.0 - P_DROP)


def declare(params, spec):
    E, H = params['embedSize'], params['rnnHiddenSize']
    C, K = params['imgFeatureSize'], params.get('commonEmbeddingSize', 512)
    assert params.get('numAttentionLayers', 1) == 1, "only the default single attention hop is built"
    spec.lstm('hist1', E, H); spec.lstm('hist2', H, H)          # mn-att:27-31
    spec.lstm('ques1', E, H); spec.lstm('ques2', H, H)          # mn-att:37-41
    spec.linear('mn1', H, H); spec.linear('mn2', H, H)          # mn-att:64-65
    spec.linear('img_proj', C, H)                               # mn-att:77
    spec.linear('img_common', H, K)                             # mn-att:84
    spec.linear('ques_common', H, K)                            # mn-att:88
    spec.linear('att', K, 1)                                    # mn-att:93
    spec.linear('out', H, H)                                    # mn-att:106


class Encoder(object):
    def __init__(self, params, fp, ws, drop, streams=None):
        self.p, self.fp, self.ws, self.drop = params, fp, ws, drop
        self.streams = streams if streams is not None else StreamPool(None, enabled=False)
        E, H = params['embedSize'], params['rnnHiddenSize']
        self.E, self.H = E, H
        self.C, self.K = params['imgFeatureSize'], params.get('commonEmbeddingSize', 512)
        self.S2 = params['imgSpatialSize'] ** 2
        self.R = params['maxQuesCount']
        self.wordEmbed = ('embed', fp.w['embed'], fp.g['embed'])      # enc.wordEmbed (mn-att:21,112)
        self.hist1, self.hist2 = SeqLSTM(fp, 'hist1', E, H, ws), SeqLSTM(fp, 'hist2', H, H, ws)
        self.ques1, self.ques2 = SeqLSTM(fp, 'ques1', E, H, ws), SeqLSTM(fp, 'ques2', H, H, ws)
        self.mn1, self.mn2 = Linear(fp, 'mn1', H, H, ws), Linear(fp, 'mn2', H, H, ws)
        self.img_proj = Linear(fp, 'img_proj', self.C, H, ws)
        self.ques_common = Linear(fp, 'ques_common', H, self.K, ws)
        self.out = Linear(fp, 'out', H, H, ws)

    def _embed(self, name, tok):
        T, N = tok.shape
        m = self.drop.mask(name + '_emb', T * N * self.E, P_DROP)
        x = self.ws.get(name + '.x', (T * N, self.E))
        ops.embed_gather(self.fp.w['embed'], tok, x, mask=m, scale=SCALE)
        return x, m

    # -- one text branch: embed -> dropout -> 2 x SeqLSTM(maskZero) -> last step (mn-att:24-45)
    def _branch_fwd(self, name, tok, l1, l2):
        T, N = tok.shape
        x, m = self._embed(name, tok)
        h1 = l1.forward(x, T, N, tok_mask=tok)
        h2 = l2.forward(h1.view(T * N, self.H), T, N, tok_mask=tok)
        return h2[T - 1], m

    def _branch_bwd(self, tok, l1, l2, dlast, m):
        dx = lstm_stack_backward([l1, l2], dh_last_top=dlast)[0]
        ops.embed_scatter_acc(self.fp.g['embed'], tok, dx, mask=m, scale=SCALE)

    def output_buffer(self, inputs):
        """the tensor forward() will return (lets the host enqueue consumers' launches first)"""
        return self.ws.get('out.y', (inputs[0].shape[1], self.H))

    def forward(self, inputs):
        ques, img, hist, mask = inputs
        ws, H, K, S2, R = self.ws, self.H, self.K, self.S2, self.R
        Tq, N = ques.shape
        B = N // R
        self.inputs, self.N, self.B = inputs, N, B
        # history branch || question branch || per-image projection (independent launch chains)
        fused = os.environ.get('VD_LSTM2', '1') != '0'
        if not fused:
            with self.streams.fork('hist'):
                self.h3, self.m_h = self._branch_fwd('h', hist, self.hist1, self.hist2)
        with self.streams.fork('img'):
            self.pre = self.img_proj.forward(img, B * S2, tanh=True)            # mn-att:74-78 (pre-dropout)
            self.m1 = self.drop.mask('img_tr', N * S2 * H, P_DROP)
            self.m2 = self.drop.mask('iqc', N * S2 * K, P_DROP)
        if fused:
            # both two-layer stacks advance together: one grouped launch per wavefront tick
            hx, self.m_h = self._embed('h', hist)
            qx, self.m_q = self._embed('q', ques)
            hT, qT = hist.shape[0], ques.shape[0]
            hh, qh = lstm2_bundle_forward([(self.hist1, self.hist2, hx, hT, N, hist),
                                           (self.ques1, self.ques2, qx, qT, N, ques)])
            self.h3, self.q3 = hh[hT - 1], qh[qT - 1]
        else:
            self.q3, self.m_q = self._branch_fwd('q', ques, self.ques1, self.ques2)
            self.streams.join('hist')
        # memory attention over the R history facts of each dialog (mn-att:48-62)
        self.prob = ws.get('mn.prob', (N, R))
        self.hatt = ws.get('mn.hatt', (N, H))
        ops.mn_attention_forward(self.q3, self.h3, mask, self.prob, self.hatt, B, R, H)
        self.m_hatt = self.drop.mask('hatt', N * H, P_DROP)
        hatt_d = dropout_forward(ws, 'mn.hatt_d', self.hatt, self.m_hatt, SCALE)
        hattTr = self.mn1.forward(hatt_d, N, tanh=True)                         # mn-att:64
        s2 = ops.axpby(hattTr, self.q3, ws.get('mn.s2', (N, H)), 1.0, 1.0)      # CAddTable
        qh2 = self.mn2.forward(s2, N, tanh=True)                                # mn-att:65
        self.qh2 = qh2
        # image attention (mn-att:68-104); `pre` is per image, dropout masks are per round
        self.streams.join('img')
        sc = SCALE if self.m1 is not None else 1.0
        self.sc = sc
        qc = self.ques_common.forward(qh2, N)                                   # mn-att:88
        self.iqc = ws.get('att.iqc', (N * S2, K))
        ops.img_common_forward(self.pre, self.m1, self.fp.w['img_common.W'], self.fp.w['img_common.b'], qc, self.m2,
                               self.iqc, N, R, S2, H, K, sc)                    # mn-att:83-92
        self.patt = ws.get('att.p', (N, S2))
        u1 = ws.get('att.u1', (N, H))
        ops.img_att_forward(self.iqc, self.fp.w['att.W'], self.fp.w['att.b'], self.pre, self.m1, qh2, self.patt, u1, N,
                            R, S2, H, K, sc)                                    # mn-att:93-102
        self.m_u = self.drop.mask('u', N * H, P_DROP)
        u1_d = dropout_forward(ws, 'att.u1_d', u1, self.m_u, SCALE)
        self.output = self.out.forward(u1_d, N, tanh=True)                      # mn-att:106
        return self.output

    def backward(self, inputs, grad_output):
        ques, img, hist, mask = inputs
        ws, H, K, S2, R, N, B, sc = self.ws, self.H, self.K, self.S2, self.R, self.N, self.B, self.sc
        G = self.fp.g
        du1d = self.out.backward(grad_output)
        du1 = dropout_backward(ws, 'att.du1', du1d, self.m_u, SCALE)           # = datt, and residual into dqh2
        dqc = ws.get('att.dqc', (N, K))
        ops.img_att_backward(self.iqc, self.fp.w['att.W'], self.pre, self.m1, self.m2, self.patt, du1, G['att.W'],
                             G['att.b'], dqc, N, R, S2, H, K, sc)               # iqc now holds dz
        dz = self.iqc
        ops.colsum_acc(dz, G['img_common.b'], M=N * S2, N=K)
        ops.img_common_wgrad(dz, self.pre, self.m1, G['img_common.W'], N, R, S2, H, K, sc)
        dpre = ws.get('att.dpre', (B * S2, H))
        dpre.zero_()
        ops.img_tr_backward(dz, self.fp.w['img_common.W'], self.patt, du1, self.m1, dpre, N, R, S2, H, K, sc)
        self.img_proj.backward(dpre, need_dx=False)                            # tanh' + dW, db of mn-att:77
        dqh2 = self.ques_common.backward(dqc)
        dqh2 = ops.axpby(dqh2, du1, ws.get('mn.dqh2', (N, H)), 1.0, 1.0)
        ds2 = self.mn2.backward(dqh2)                                          # -> d(hAttTr) and d(q3)
        dhatt_d = self.mn1.backward(ds2)
        dhatt = dropout_backward(ws, 'mn.dhatt', dhatt_d, self.m_hatt, SCALE)
        dq_att = ws.get('mn.dq', (N, H))
        dh3 = ws.get('mn.dh', (N, H))
        ops.mn_attention_backward(self.q3, self.h3, self.prob, dhatt, dq_att, dh3, B, R, H)
        dq3 = ops.axpby(dq_att, ds2, ws.get('mn.dq3', (N, H)), 1.0, 1.0)
        if os.environ.get('VD_LSTM2', '1') != '0':
            dhx, dqx = lstm2_bundle_backward([(self.hist1, self.hist2, dh3), (self.ques1, self.ques2, dq3)])
            ops.embed_scatter_acc(self.fp.g['embed'], hist, dhx, mask=self.m_h, scale=SCALE)
            ops.embed_scatter_acc(self.fp.g['embed'], ques, dqx, mask=self.m_q, scale=SCALE)
            return None
        with self.streams.fork('hist'):
            self._branch_bwd(hist, self.hist1, self.hist2, dh3, self.m_h)
        self._branch_bwd(ques, self.ques1, self.ques2, dq3, self.m_q)
        self.streams.join('hist')
        return None


def model(params, fp, ws, drop, streams=None):
    return Encoder(params, fp, ws, drop, streams)
