"""Hierarchical recurrent encoder with image: counterpart of encoders/hre-ques-im-hist.lua:5-97.

  hist -> embed -> L x SeqLSTM(maskZero) -> last step                                  (:26-41)
  img  -> Linear(F, imgEmbedSize) -> MaskTime (replicate over the question's time steps,
          zero where the question token is 0)                                          (:43-53)
  [word embedding ; masked image embedding] -> L x SeqLSTM(maskZero) -> last step      (:66-79)
  [ques state ; hist state] viewed [B x R x 2H], transposed, dialog-level SeqLSTM(2H, H)
          over the R rounds, transposed back                                           (:85-94)
The JoinTables in front of the two LSTMs are folded into split input projections (no concat copy).
Inputs: ques [Tq x N], img [B x F], hist [Th x N].  No dropout layer is instantiated in this file."""
import numpy as np
import torch

from .. import ops
from ..nn import SeqLSTM, Linear, StreamPool, lstm_stack_forward, lstm_stack_backward


def declare(params, spec):
    E, H, F, DI = params['embedSize'], params['rnnHiddenSize'], params['imgFeatureSize'], params['imgEmbedSize']
    for l in range(params['numLayers']):
        spec.lstm('hist%d' % (l + 1), E if l == 0 else H, H)
    spec.linear('img_embed', F, DI)
    for l in range(params['numLayers']):
        spec.lstm('ques%d' % (l + 1), (E + DI) if l == 0 else H, H)
    spec.lstm('dialog', 2 * H, H)


class Encoder(object):
    def __init__(self, params, fp, ws, drop, streams=None):
        self.p, self.fp, self.ws, self.drop = params, fp, ws, drop
        self.streams = streams if streams is not None else StreamPool(None, enabled=False)
        E, H, F, DI = params['embedSize'], params['rnnHiddenSize'], params['imgFeatureSize'], params['imgEmbedSize']
        self.E, self.H, self.F, self.DI, self.R = E, H, F, DI, params['maxQuesCount']
        self.wordEmbed = ('embed', fp.w['embed'], fp.g['embed'])
        L = params['numLayers']
        self.histLayers = [SeqLSTM(fp, 'hist%d' % (l + 1), E if l == 0 else H, H, ws) for l in range(L)]
        self.img_embed = Linear(fp, 'img_embed', F, DI, ws)
        self.rnnLayers = [SeqLSTM(fp, 'ques%d' % (l + 1), (E + DI) if l == 0 else H, H, ws,
                                  part_dims=[E, DI] if l == 0 else None) for l in range(L)]
        self.dialog = SeqLSTM(fp, 'dialog', 2 * H, H, ws, part_dims=[H, H])
        self._idx = {}

    def output_buffer(self, inputs):
        return self.ws.get('hre.out', (inputs[0].shape[1], self.H))

    def _indices(self, N, device):
        """round-major <-> dialog-major row permutations (nn.View + nn.Transpose({1,2}), :88-93)"""
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

    def forward(self, inputs):
        ques, img, hist = inputs
        ws, H, E, DI, R = self.ws, self.H, self.E, self.DI, self.R
        Tq, N = ques.shape
        Th = hist.shape[0]
        B = N // R
        rep, to_rb, to_n = self._indices(N, ques.device)
        with self.streams.fork('hist'):
            hx = ws.get('h.x', (Th * N, E))
            ops.embed_gather(self.fp.w['embed'], hist, hx)
            hh = lstm_stack_forward(self.histLayers, hx, Th, N, hist)[Th - 1]
            hh_rb = ops.embed_gather(hh, to_rb, ws.get('hre.hh_rb', (N, H)))
        img_rep = ops.embed_gather(img, rep, ws.get('img.rep', (N, self.F)))
        imgE = self.img_embed.forward(img_rep, N)                                   # :43-48
        xi = ops.mask_time_forward(imgE, ques, ws.get('q.ximg', (Tq * N, DI)), Tq, N, DI)   # MaskTime :50-53
        qx = ws.get('q.x', (Tq * N, E))
        ops.embed_gather(self.fp.w['embed'], ques, qx)
        hq = lstm_stack_forward(self.rnnLayers, [qx, xi], Tq, N, ques)[Tq - 1]
        hq_rb = ops.embed_gather(hq, to_rb, ws.get('hre.hq_rb', (N, H)))
        self.streams.join('hist')
        d = self.dialog.forward([hq_rb, hh_rb], R, B, None)                         # :90-94 (no maskZero)
        self.output = ops.embed_gather(d.view(N, H), to_n, ws.get('hre.out', (N, H)))
        return self.output

    def backward(self, inputs, grad_output):
        ques, img, hist = inputs
        ws, H, E, DI, R = self.ws, self.H, self.E, self.DI, self.R
        Tq, N = ques.shape
        B = N // R
        rep, to_rb, to_n = self._indices(N, ques.device)
        g_rb = ops.embed_gather(grad_output, to_rb, ws.get('hre.g_rb', (N, H)))
        dq_rb, dh_rb = self.dialog.backward(dh_seq=g_rb.view(R, B, H))
        dq = ops.embed_gather(dq_rb, to_n, ws.get('hre.dq', (N, H)))
        dh = ops.embed_gather(dh_rb, to_n, ws.get('hre.dh', (N, H)))
        with self.streams.fork('hist'):
            dhx = lstm_stack_backward(self.histLayers, dh_last_top=dh)[0]
            ops.embed_scatter_acc(self.fp.g['embed'], hist, dhx)
        dqx, dxi = lstm_stack_backward(self.rnnLayers, dh_last_top=dq)
        ops.embed_scatter_acc(self.fp.g['embed'], ques, dqx)
        dimgE = ops.mask_time_backward(dxi, ques, ws.get('q.dimgE', (N, DI)), Tq, N, DI)
        self.img_embed.backward(dimgE, need_dx=False)
        self.streams.join('hist')


def model(params, fp, ws, drop, streams=None):
    return Encoder(params, fp, ws, drop, streams)
