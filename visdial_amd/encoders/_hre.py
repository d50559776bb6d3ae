"""Hierarchical recurrent encoders (encoders/hre-ques-hist.lua, hre-ques-im-hist.lua:5-97,
hrea-ques-im-hist.lua):

  hist -> embed -> L x SeqLSTM(maskZero) -> last step
  [word embedding ; MaskTime(Linear(F, imgEmbedSize)(img))] -> L x SeqLSTM(maskZero) -> last step
        (image part only for *-im-*; hrea puts Dropout(0.5) in front of the image Linear, hrea:47)
  hrea only: additive attention of every question over the history states of rounds <= its own
        (two Linear(H,1), MaskFuture, ReplaceZero(-inf), SoftMax; hrea:83-131) replaces the history state
  [first ; second] viewed [B x R x 2H], transposed, dialog-level SeqLSTM(2H, H) over the R rounds
The JoinTables in front of the LSTMs are folded into split input projections (no concat copy)."""
import numpy as np
import torch

from .. import ops
from ..nn import SeqLSTM, Linear, StreamPool, dropout_forward, dropout_backward, lstm_stack_forward, lstm_stack_backward


def make(use_im, attention):
    def declare(params, spec):
        E, H, F = params['embedSize'], params['rnnHiddenSize'], params['imgFeatureSize']
        DI = params['imgEmbedSize'] if use_im else 0
        for l in range(params['numLayers']):
            spec.lstm('hist%d' % (l + 1), E if l == 0 else H, H)
        if use_im:
            spec.linear('img_embed', F, DI)
        for l in range(params['numLayers']):
            spec.lstm('ques%d' % (l + 1), (E + DI) if l == 0 else H, H)
        if attention:
            spec.linear('att_q', H, 1)
            spec.linear('att_h', H, 1)
        spec.lstm('dialog', 2 * H, H)

    class Encoder(object):
        def __init__(self, params, fp, ws, drop, streams=None):
            self.p, self.fp, self.ws, self.drop = params, fp, ws, drop
            self.streams = streams if streams is not None else StreamPool(None, enabled=False)
            E, H, F = params['embedSize'], params['rnnHiddenSize'], params['imgFeatureSize']
            DI = params['imgEmbedSize'] if use_im else 0
            self.E, self.H, self.F, self.DI, self.R = E, H, F, DI, params['maxQuesCount']
            self.wordEmbed = ('embed', fp.w['embed'], fp.g['embed'])
            L = params['numLayers']
            self.histLayers = [SeqLSTM(fp, 'hist%d' % (l + 1), E if l == 0 else H, H, ws) for l in range(L)]
            if use_im:
                self.img_embed = Linear(fp, 'img_embed', F, DI, ws)
            self.rnnLayers = [SeqLSTM(fp, 'ques%d' % (l + 1), (E + DI) if l == 0 else H, H, ws,
                                      part_dims=[E, DI] if (l == 0 and use_im) else None) for l in range(L)]
            self.dialog = SeqLSTM(fp, 'dialog', 2 * H, H, ws, part_dims=[H, H])
            if attention:
                assert self.R <= 16
            self._idx = {}

        def output_buffer(self, inputs):
            return self.ws.get('hre.out', (inputs[0].shape[1], self.H))

        def _indices(self, N, device):
            """round-major <-> dialog-major row permutations (nn.View + nn.Transpose({1,2}), hre:88-93)"""
            ix = self._idx.get(N)
            if ix is None:
                R = self.R
                B = N // R
                n = np.arange(N)
                rep = (n // R).astype(np.int32)                               # image row of QA round n
                to_rb = ((n % B) * R + n // B).astype(np.int32)               # row r*B+b  <- n = b*R + r
                to_n = ((n % R) * B + n // R).astype(np.int32)                # row b*R+r  <- r*B + b
                ix = tuple(torch.from_numpy(a).to(device) for a in (rep, to_rb, to_n))
                self._idx[N] = ix
            return ix

        def _unpack(self, inputs):
            it = iter(inputs)
            ques = next(it)
            img = next(it) if use_im else None
            return ques, img, next(it)

        def forward(self, inputs):
            ques, img, hist = self._unpack(inputs)
            fp, ws, H, E, DI, R = self.fp, self.ws, self.H, self.E, self.DI, self.R
            Tq, N = ques.shape
            Th = hist.shape[0]
            B = N // R
            rep, to_rb, to_n = self._indices(N, ques.device)
            with self.streams.fork('hist'):
                hx = ws.get('h.x', (Th * N, E))
                ops.embed_gather(fp.w['embed'], hist, hx)
                self.hh = lstm_stack_forward(self.histLayers, hx, Th, N, hist)[Th - 1]
            qx = ws.get('q.x', (Tq * N, E))
            ops.embed_gather(fp.w['embed'], ques, qx)
            x = qx
            if use_im:
                img_rep = ops.embed_gather(img, rep, ws.get('img.rep', (N, self.F)))
                self.m_img = self.drop.mask('img', N * self.F, 0.5) if attention else None    # hrea:47
                img_in = dropout_forward(ws, 'img.in', img_rep, self.m_img, 2.0)
                imgE = self.img_embed.forward(img_in, N)                                      # hre:43-48
                xi = ops.mask_time_forward(imgE, ques, ws.get('q.ximg', (Tq * N, DI)), Tq, N, DI)   # MaskTime hre:50-53
                x = [qx, xi]
            self.hq = lstm_stack_forward(self.rnnLayers, x, Tq, N, ques)[Tq - 1]
            self.streams.join('hist')
            first, second = self.hq, self.hh
            if attention:                                                                      # hrea:83-131
                self.sq = ops.rowdot_forward(self.hq, fp.w['att_q.W'], fp.w['att_q.b'], ws.get('hrea.sq', (N,)), N, H)
                self.sh = ops.rowdot_forward(self.hh, fp.w['att_h.W'], fp.w['att_h.b'], ws.get('hrea.sh', (N,)), N, H)
                self.P = ws.get('hrea.P', (N, R))
                att = ws.get('hrea.att', (N, H))
                ops.hrea_attention_forward(self.sq, self.sh, self.hh, self.P, att, B, R, H)
                first, second = att, self.hq                                                   # concat4: {att, ques}
            f_rb = ops.embed_gather(first, to_rb, ws.get('hre.f_rb', (N, H)))
            s_rb = ops.embed_gather(second, to_rb, ws.get('hre.s_rb', (N, H)))
            d = self.dialog.forward([f_rb, s_rb], R, B, None)                                  # hre:90-94 (no maskZero)
            self.output = ops.embed_gather(d.view(N, H), to_n, ws.get('hre.out', (N, H)))
            return self.output

        def backward(self, inputs, grad_output):
            ques, img, hist = self._unpack(inputs)
            fp, ws, H, E, DI, R = self.fp, self.ws, self.H, self.E, self.DI, self.R
            G = fp.g
            Tq, N = ques.shape
            B = N // R
            rep, to_rb, to_n = self._indices(N, ques.device)
            g_rb = ops.embed_gather(grad_output, to_rb, ws.get('hre.g_rb', (N, H)))
            df_rb, ds_rb = self.dialog.backward(dh_seq=g_rb.view(R, B, H))
            dfirst = ops.embed_gather(df_rb, to_n, ws.get('hre.dfirst', (N, H)))
            dsecond = ops.embed_gather(ds_rb, to_n, ws.get('hre.dsecond', (N, H)))
            if attention:
                dsq, dsh = ws.get('hrea.dsq', (N,)), ws.get('hrea.dsh', (N,))
                dh_att = ws.get('hrea.dh', (N, H))
                ops.hrea_attention_backward(self.hh, self.P, dfirst, dsq, dsh, dh_att, B, R, H)
                dq_s = ops.rowdot_backward(self.hq, fp.w['att_q.W'], dsq, G['att_q.W'], G['att_q.b'],
                                           ws.get('hrea.dq_s', (N, H)), N, H)
                dh_s = ops.rowdot_backward(self.hh, fp.w['att_h.W'], dsh, G['att_h.W'], G['att_h.b'],
                                           ws.get('hrea.dh_s', (N, H)), N, H)
                dq = ops.axpby(dsecond, dq_s, ws.get('hrea.dq', (N, H)), 1.0, 1.0)
                dh = ops.axpby(dh_att, dh_s, ws.get('hrea.dhh', (N, H)), 1.0, 1.0)
            else:
                dq, dh = dfirst, dsecond
            with self.streams.fork('hist'):
                dhx = lstm_stack_backward(self.histLayers, dh_last_top=dh)[0]
                ops.embed_scatter_acc(G['embed'], hist, dhx)
            dxs = lstm_stack_backward(self.rnnLayers, dh_last_top=dq)
            ops.embed_scatter_acc(G['embed'], ques, dxs[0])
            if use_im:
                dimgE = ops.mask_time_backward(dxs[1], ques, ws.get('q.dimgE', (N, DI)), Tq, N, DI)
                self.img_embed.backward(dimgE, need_dx=False)
            self.streams.join('hist')

    def model(params, fp, ws, drop, streams=None):
        return Encoder(params, fp, ws, drop, streams)

    return declare, model
