"""Generative decoder: counterpart of decoders/gen.lua:3-68.

  answer_in -> shared embedding -> numLayers x SeqLSTM(maskZero) -> Linear(H, V) -> LogSoftMax,
  criterion = sum over non-pad steps of -log p(answer_out)   (model.lua:32-36,306-324)
The encoder's per-layer final (h, c) seed the decoder layers and the encoder output replaces the top
layer's initial h (forwardConnect, gen.lua:30-42); backwardConnect hands the gradients w.r.t. those
initial states back (gen.lua:45-60).  Vocabulary projection, log-softmax and NLL (+ their backward)
are one MFMA GEMM + one wave-reduction kernel that turns the logits into their gradient in place."""
import numpy as np

from .. import ops
from ..nn import SeqLSTM, lstm_stack_forward, lstm_stack_backward


def declare(params, spec):
    E, H, V = params['embedSize'], params['rnnHiddenSize'], params['vocabSize']
    for l in range(params['numLayers']):
        spec.lstm('dec%d' % (l + 1), E if l == 0 else H, H)          # gen.lua:17-22 (ansHiddenSize is never set)
    spec.linear('vocab', H, V)                                       # gen.lua:23


class Decoder(object):
    def __init__(self, params, enc, fp, ws, drop):
        self.p, self.fp, self.ws = params, fp, ws
        self.E, self.H, self.V = params['embedSize'], params['rnnHiddenSize'], params['vocabSize']
        self.Vp = (self.V + 3) // 4 * 4
        self.emb, self.demb = enc.wordEmbed[1], enc.wordEmbed[2]     # gen.lua:10-11
        L = params['numLayers']
        self.rnnLayers = [SeqLSTM(fp, 'dec%d' % (l + 1), self.E if l == 0 else self.H, self.H, ws) for l in range(L)]
        self.Wv, self.bv = fp.w['vocab.W'], fp.w['vocab.b']
        self.dWv, self.dbv = fp.g['vocab.W'], fp.g['vocab.b']

    def forward_backward_gen(self, model, inputs, dec_in, encOut, onlyForward):
        """decoder forward, criterion, and (unless onlyForward) decoder + encoder backward
        (model.lua:306-324).  Returns the summed NLL."""
        ws, H, V, Vp = self.ws, self.H, self.V, self.Vp
        ain, aout = dec_in['answer_in'], dec_in['answer_out']
        Ta, N = ain.shape
        rows = Ta * N
        x = ws.get('dec.x', (rows, self.E))
        ops.embed_gather(self.emb, ain, x)
        h = lstm_stack_forward(self.rnnLayers, x, Ta, N, ain).view(rows, H)
        logits = ws.get('dec.logits', (rows, Vp))
        ops.gemm_nt(h, self.Wv, logits, bias=self.bv, M=rows, N=V, K=H, ldc=Vp)
        loss_rows = ws.get('dec.loss_rows', (rows,))
        ops.logsoftmax_nll(logits, V, ain.view(-1), aout.view(-1), loss_rows, write_grad=not onlyForward)
        if not onlyForward:
            dlog = logits                                                      # now d loss / d logits
            ops.gemm_tn_acc(dlog, h, self.dWv, M=V, N=H, K=rows, lda=Vp)
            ops.colsum_acc(dlog, self.dbv, M=rows, N=V, ld=Vp)
            dh = ws.get('dec.dh', (rows, H))
            ops.gemm_nn(dlog, self.Wv, dh, M=rows, N=H, K=V, lda=Vp)
            dx = lstm_stack_backward(self.rnnLayers, dh_seq_top=dh.view(Ta, N, H))[0]
            ops.embed_scatter_acc(self.demb, ain, dx)
            gradDecOut = backwardConnect(model.encoder, self)
            model.encoder.backward(inputs, gradDecOut)
        return float(loss_rows.cpu().numpy().astype(np.float64).sum())


def model(params, enc, fp, ws, drop):
    return Decoder(params, enc, fp, ws, drop)


def forwardConnect(enc, dec, encOut, seqLen):
    """gen.lua:30-42"""
    layers = getattr(enc, 'rnnLayers', None)
    if layers is not None:
        for ii in range(len(layers)):
            dec.rnnLayers[ii].userPrevOutput = layers[ii].output[seqLen - 1]
            dec.rnnLayers[ii].userPrevCell = layers[ii].cell[seqLen - 1]
        dec.rnnLayers[len(layers) - 1].userPrevOutput = encOut
    else:
        dec.rnnLayers[-1].userPrevOutput = encOut


def backwardConnect(enc, dec):
    """gen.lua:45-60: returns dL/d encOut"""
    layers = getattr(enc, 'rnnLayers', None)
    if layers is not None:
        n = len(dec.rnnLayers)
        for ii in range(n):
            layers[ii].userNextGradCell = dec.rnnLayers[ii].userGradPrevCell
            if ii != n - 1:
                layers[ii].gradPrevOutput = dec.rnnLayers[ii].userGradPrevOutput
        return dec.rnnLayers[len(layers) - 1].userGradPrevOutput
    return dec.rnnLayers[-1].userGradPrevOutput


def decoderConnect(dec):
    """gen.lua:63-68 (sampling: chain the decoder to itself)"""
    for l in dec.rnnLayers:
        l.userPrevCell = l.cell[0]
        l.userPrevOutput = l.output[0]
