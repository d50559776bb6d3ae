"""Late-fusion encoders (encoders/lf-ques.lua, lf-ques-im.lua, lf-ques-hist.lua, lf-ques-im-hist.lua):
  [quesLSTM last step ; img ; histLSTM last step] -> Dropout(p) -> Linear(., H) -> Tanh
with the image / history parts present according to the encoder NAME (opts.lua:54-59).  History is the
concatenated dialog (opts.lua:59).  Exposes wordEmbed, rnnLayers (read by decoders/gen.lua:30-60) and
histLayers.  Inputs: ques [Tq x N], [img [B x F]], [hist [Th x N]]."""
from .. import ops
from ..nn import SeqLSTM, StreamPool, lstm_stack_forward, lstm_stack_backward
from ._blocks import CatLinear, round_index


def make(use_im, use_hist):
    def declare(params, spec):
        E, H, F = params['embedSize'], params['rnnHiddenSize'], params['imgFeatureSize']
        for l in range(params['numLayers']):
            spec.lstm('ques%d' % (l + 1), E if l == 0 else H, H)        # lf-ques-im-hist.lua:19-26
        if use_hist:
            for l in range(params['numLayers']):
                spec.lstm('hist%d' % (l + 1), E if l == 0 else H, H)    # :38-45
        spec.linear('fuse', H * (2 if use_hist else 1) + (F if use_im else 0), H)   # :58

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
            self.histLayers = [SeqLSTM(fp, 'hist%d' % (l + 1), E if l == 0 else H, H, ws) for l in range(L)] if use_hist else None
            dims = [H] + ([F] if use_im else []) + ([H] if use_hist else [])
            self.fuse = CatLinear(fp, 'fuse', dims, H, ws)
            self._idx = {}

        def output_buffer(self, inputs):
            return self.ws.get('fuse.y', (inputs[0].shape[1], self.H))

        def _unpack(self, inputs):
            it = iter(inputs)
            ques = next(it)
            img = next(it) if use_im else None
            hist = next(it) if use_hist else None
            return ques, img, hist

        def forward(self, inputs):
            ques, img, hist = self._unpack(inputs)
            ws = self.ws
            Tq, N = ques.shape
            hh_last = None
            if use_hist:
                with self.streams.fork('hist'):
                    Th = hist.shape[0]
                    hx = ws.get('h.x', (Th * N, self.E))
                    ops.embed_gather(self.fp.w['embed'], hist, hx)
                    hh_last = lstm_stack_forward(self.histLayers, hx, Th, N, hist)[Th - 1]
            qx = ws.get('q.x', (Tq * N, self.E))
            ops.embed_gather(self.fp.w['embed'], ques, qx)
            parts = [lstm_stack_forward(self.rnnLayers, qx, Tq, N, ques)[Tq - 1]]
            if use_im:
                parts.append(ops.embed_gather(img, round_index(N, self.R, ques.device, self._idx),
                                              ws.get('img.rep', (N, self.F))))        # repeat per round
            if use_hist:
                self.streams.join('hist')
                parts.append(hh_last)
            scale = 1.0 / (1.0 - self.pdrop) if self.pdrop > 0 else 1.0
            D = self.fuse.D
            m = self.drop.mask('fuse', N * D, self.pdrop)
            self.output = self.fuse.forward(parts, N, mask=m, scale=scale)
            return self.output

        def backward(self, inputs, grad_output):
            ques, img, hist = self._unpack(inputs)
            need = [True] + ([False] if use_im else []) + ([True] if use_hist else [])
            grads = self.fuse.backward(grad_output, need=need)
            if use_hist:
                with self.streams.fork('hist'):
                    dhx = lstm_stack_backward(self.histLayers, dh_last_top=grads[-1])[0]
                    ops.embed_scatter_acc(self.fp.g['embed'], hist, dhx)
            dqx = lstm_stack_backward(self.rnnLayers, dh_last_top=grads[0])[0]
            ops.embed_scatter_acc(self.fp.g['embed'], ques, dqx)
            if use_hist:
                self.streams.join('hist')

    def model(params, fp, ws, drop, streams=None):
        return Encoder(params, fp, ws, drop, streams)

    return declare, model
