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


    def step_logprobs(self, tok, hidden):
        """One decoder step for K hypotheses (the `self.decoder:forward(beams[{{step-1}}])` of model.lua:518):
        tok device int32 [1 x K]; hidden = per-layer (h [K x H], c [K x H]).  Returns (log-probs [K x V] on the
        host, new per-layer (h, c) device tensors)."""
        ws, H, V, Vp = self.ws, self.H, self.V, self.Vp
        K = tok.shape[1]
        for l, (h0, c0) in zip(self.rnnLayers, hidden):
            l.userPrevOutput, l.userPrevCell = h0, c0
        x = ws.get('gen1.x', (K, self.E))
        ops.embed_gather(self.emb, tok, x)
        h = lstm_stack_forward(self.rnnLayers, x, 1, K, tok).view(K, H)
        logits = ws.get('gen1.logits', (K, Vp))
        ops.gemm_nt(h, self.Wv, logits, bias=self.bv, M=K, N=V, K=H, ldc=Vp)
        ops.log_softmax_rows(logits, V)
        new_hidden = [(l.output[0].clone(), l.cell[0].clone()) for l in self.rnnLayers]
        logp = logits[:, :V].cpu().numpy()
        logp[tok[0].cpu().numpy() == 0] = 0.0        # MaskZero(Linear) + MaskZero(LogSoftMax) (gen.lua:23-24): a pad token's row is all zeros
        return logp, new_hidden

    def retrieve_lhood(self, model, option_in, option_out, encOut, seqLen):
        """Model:retrieveBatch gen branch (model.lua:392-420) + utils.computeLhood (utils.lua:86-102).
        The reference loops over the 100 options; here chunks of options are ONE decoder batch (rows =
        round x option), seeded by the replicated encoder state, and the [rows x V] logits only ever exist
        for one chunk.  option_in/out: [T x N*O] int32 time-major, row = n*O + o.  Returns lhood [N x O]."""
        import torch
        ws, H, V, Vp = self.ws, self.H, self.V, self.Vp
        T, NO = option_in.shape
        N = encOut.shape[0]
        O = NO // N
        enc = model.encoder
        layers = getattr(enc, 'rnnLayers', None)
        lhood = ws.get('ret.lhood', (N, O))
        oc = max(1, min(O, int((1 << 30) // max(1, T * N * Vp))))           # options per chunk: <= 4 GiB of logits
        tin = option_in.view(T, N, O)
        tout = option_out.view(T, N, O)
        for o0 in range(0, O, oc):
            o1 = min(O, o0 + oc)
            C = o1 - o0
            rows = N * C
            cin = tin[:, :, o0:o1].contiguous().view(T, rows)               # plumbing: strided int copies
            cout = tout[:, :, o0:o1].contiguous().view(T, rows)
            idx = ws.get('ret.idx', (rows,), torch.int32)
            idx.copy_(torch.arange(rows, device=idx.device, dtype=torch.int32) // C)
            # forwardConnect (gen.lua:30-42) with the encoder state replicated over the chunk's options
            def rep(x, key):
                return ops.embed_gather(x, idx, ws.get(key, (rows, H)))
            if layers is not None:
                for ii in range(len(layers)):
                    self.rnnLayers[ii].userPrevOutput = rep(layers[ii].output[seqLen - 1], 'ret.h0_%d' % ii)
                    self.rnnLayers[ii].userPrevCell = rep(layers[ii].cell[seqLen - 1], 'ret.c0_%d' % ii)
                self.rnnLayers[len(layers) - 1].userPrevOutput = rep(encOut, 'ret.enc')
            else:
                self.rnnLayers[-1].userPrevOutput = rep(encOut, 'ret.enc')
            x = ws.get('ret.x', (T * rows, self.E))
            ops.embed_gather(self.emb, cin, x)
            h = lstm_stack_forward(self.rnnLayers, x, T, rows, cin).view(T * rows, H)
            logits = ws.get('ret.logits', (T * rows, Vp))
            ops.gemm_nt(h, self.Wv, logits, bias=self.bv, M=T * rows, N=V, K=H, ldc=Vp)
            nll = ws.get('ret.nll', (T, rows))
            ops.logsoftmax_nll(logits, V, cin.view(-1), cout.view(-1), nll.view(-1), write_grad=False)
            acc = ws.get('ret.acc', (rows,))
            ops.zero(acc)
            ops.colsum_acc(nll, acc, M=T, N=rows)                            # sum over time (utils.lua:98)
            ops.copy_2d(lhood, O, acc, C, N, C, dst_off=o0)
        return ops.axpby(lhood, None, lhood, -1.0, 0.0)                       # log-likelihood = -NLL


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
