"""Late-fusion question + image + history encoder: counterpart of encoders/lf-ques-im-hist.lua:3-62.
[quesLSTM last step ; img ; histLSTM last step] -> Dropout(p) -> Linear(2H+F, H) -> Tanh.  History is the
concatenated dialog (opts.lua:59).  Inputs: ques [Tq x N], img [B x F] (the reference's per-round repeat,
model.lua:267-269, is a row gather here), hist [Th x N]."""
import numpy as np
import torch

from .. import ops
from ..nn import (SeqLSTM, Linear, StreamPool, dropout_forward, dropout_backward, lstm_stack_forward,
                  lstm_stack_backward)


def declare(params, spec):
    E, H, F = params['embedSize'], params['rnnHiddenSize'], params['imgFeatureSize']
    for l in range(params['numLayers']):
        spec.lstm('ques%d' % (l + 1), E if l == 0 else H, H)        # lf-ques-im-hist.lua:19-26
    for l in range(params['numLayers']):
        spec.lstm('hist%d' % (l + 1), E if l == 0 else H, H)        # :38-45
    spec.linear('fuse', 2 * H + F, H)                               # :58


class Encoder(object):
    def __init__(self, params, fp, ws, drop, streams=None):
        self.p, self.fp, self.ws, self.drop = params, fp, ws, drop
        self.streams = streams if streams is not None else StreamPool(None, enabled=False)
        E, H, F = params['embedSize'], params['rnnHiddenSize'], params['imgFeatureSize']
        self.E, self.H, self.F, self.R = E, H, F, params['maxQuesCount']
        self.pdrop = float(params.get('dropout', 0.5))
        self.wordEmbed = ('embed', fp.w['embed'], fp.g['embed'])
        L = params['numLayers']
        self.rnnLayers = [SeqLSTM(fp, 'ques%d' % (l + 1), E if l == 0 else H, H, ws) for l in range(L)]
        self.histLayers = [SeqLSTM(fp, 'hist%d' % (l + 1), E if l == 0 else H, H, ws) for l in range(L)]
        self.fuse = Linear(fp, 'fuse', 2 * H + F, H, ws)
        self._idx = {}

    def output_buffer(self, inputs):
        return self.ws.get('fuse.y', (inputs[0].shape[1], self.H))

    def _round_index(self, N, device):
        idx = self._idx.get(N)
        if idx is None:
            idx = torch.from_numpy((np.arange(N) // self.R).astype(np.int32)).to(device)
            self._idx[N] = idx
        return idx

    def forward(self, inputs):
        ques, img, hist = inputs
        ws, H, F = self.ws, self.H, self.F
        Tq, N = ques.shape
        Th = hist.shape[0]
        D = 2 * H + F
        cat = ws.get('fuse.cat', (N, D))
        with self.streams.fork('hist'):
            hx = ws.get('h.x', (Th * N, self.E))
            ops.embed_gather(self.fp.w['embed'], hist, hx)
            hh = lstm_stack_forward(self.histLayers, hx, Th, N, hist)
            ops.copy_2d(cat, D, hh[Th - 1], H, N, H, dst_off=H + F)
        qx = ws.get('q.x', (Tq * N, self.E))
        ops.embed_gather(self.fp.w['embed'], ques, qx)
        hq = lstm_stack_forward(self.rnnLayers, qx, Tq, N, ques)
        ops.copy_2d(cat, D, hq[Tq - 1], H, N, H, dst_off=0)
        img_rep = ws.get('img.rep', (N, F))
        ops.embed_gather(img, self._round_index(N, ques.device), img_rep)             # repeat per round
        ops.copy_2d(cat, D, img_rep, F, N, F, dst_off=H)
        self.streams.join('hist')
        self.scale = 1.0 / (1.0 - self.pdrop) if self.pdrop > 0 else 1.0
        self.m = self.drop.mask('fuse', N * D, self.pdrop)
        catd = dropout_forward(ws, 'fuse.in', cat, self.m, self.scale)
        self.output = self.fuse.forward(catd, N, tanh=True)
        return self.output

    def backward(self, inputs, grad_output):
        ques, img, hist = inputs
        ws, H, F = self.ws, self.H, self.F
        N = ques.shape[1]
        D = 2 * H + F
        dcatd = self.fuse.backward(grad_output)
        dcat = dropout_backward(ws, 'fuse.din', dcatd, self.m, self.scale)
        dq = ws.get('fuse.dq', (N, H))
        dh = ws.get('fuse.dh', (N, H))
        ops.copy_2d(dq, H, dcat, D, N, H, src_off=0)
        ops.copy_2d(dh, H, dcat, D, N, H, src_off=H + F)
        with self.streams.fork('hist'):
            dhx = lstm_stack_backward(self.histLayers, dh_last_top=dh)[0]
            ops.embed_scatter_acc(self.fp.g['embed'], hist, dhx)
        dqx = lstm_stack_backward(self.rnnLayers, dh_last_top=dq)[0]
        ops.embed_scatter_acc(self.fp.g['embed'], ques, dqx)
        self.streams.join('hist')


def model(params, fp, ws, drop, streams=None):
    return Encoder(params, fp, ws, drop, streams)
