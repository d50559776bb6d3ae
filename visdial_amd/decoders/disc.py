"""Discriminative decoder: counterpart of decoders/disc.lua:3-38.

The reference builds 100 weight-shared (embed -> SeqLSTM(batchfirst, no maskZero) -> last step)
clones under nn.Concat(2) and scores them against the encoder output with nn.MM.  Here the 100
clones ARE one batch of N*100 option sequences: the option LSTM runs as a single [N*O x H] x
[H x 4H] fp32-MFMA recurrence whose input projection is gathered from the table Emb*Wx+b
(exact: no dropout on option embeddings, disc.lua:12-14), and scoring + cross-entropy + their
gradients are one wave-reduction kernel.
"""

import torch

from .. import ops
from ..criterion import DiscDecoderOutput
from ..nn import StreamPool


def declare(params, spec):
    spec.lstm('opt', params['embedSize'], params['rnnHiddenSize'])       # disc.lua:4


class Decoder(object):
    def __init__(self, params, enc, fp, ws, drop):
        self.p, self.fp, self.ws = params, fp, ws
        self.E, self.H, self.V = params['embedSize'], params['rnnHiddenSize'], params['vocabSize']
        self.emb, self.demb = enc.wordEmbed[1], enc.wordEmbed[2]         # shared table (disc.lua:12)
        W, dW = fp.w['opt.W'], fp.g['opt.W']
        self.Wx, self.Wh, self.b = W[:self.E], W[self.E:], fp.w['opt.b']
        self.dWx, self.dWh, self.db = dW[:self.E], dW[self.E:], fp.g['opt.b']
        self.streams = StreamPool(ws.device, enabled=False)     # Model attaches its own pool
        # arithmetic of the option recurrence (opts.py): default = the exact 3 x bf16 split (fp32-grade), 'fp32' = v_mfma_f32, 'bf16' = configs[4]
        self.flags = ops.PRECISION_FLAGS[params.get('lstmPrecision', 'split9')]

    def forward(self, inputs):
        """inputs = {options [To x N*O] int32 time-major, encOut [N x H]} -> scores [N x O]"""
        otok, enc_out = inputs
        ws, H, V = self.ws, self.H, self.V
        To, NO = otok.shape            # NO = rows the option LSTM executes: N*O, or the DISTINCT candidates (vd_uid set)
        N = enc_out.shape[0]
        self.uid = getattr(otok, 'vd_uid', None)
        O = (otok.vd_total if self.uid is not None else NO) // N
        self.To, self.NO, self.N, self.O = To, NO, N, O
        self.table = ws.get('opt.table', (V + 1, 4 * H))
        ops.gemm_nn(self.emb, self.Wx, self.table, bias=self.b, M=V + 1, N=4 * H, K=self.E)
        self.gates = ws.get('opt.gates', (To, NO, 4 * H))
        self.h = ws.get('opt.h', (To, NO, H))
        self.c = ws.get('opt.c', (To, NO, H))
        t0 = ops.prof_begin('opt_lstm_fwd')
        ops.lstm_forward(self.table, self.Wh, self.gates, self.h, self.c, To, NO, H, 0, 4 * H, tok_gather=otok,
                         flags=self.flags)
        ops.prof_end('opt_lstm_fwd', t0, 1)      # one persistent launch for all To steps
        self.optH = self.h[To - 1]
        if self.uid is not None:       # candidate (n, o) reads the state of its distinct row
            full = ws.get('opt.h_full', (N * O, H))
            ops.embed_gather(self.optH, self.uid, full)
            self.optH = full
        self.output = ws.get('opt.scores', (N, O))   # filled by the criterion call (scores + loss are one kernel)
        return DiscDecoderOutput(self.optH, enc_out, self.output, N, O, H)

    def backward(self, inputs, gradOutput):
        """decoder:backward({options, encOut}, gradCriterionOut) (model.lua:335): gradOutput = (d optH [N*O x H],
        d encOut [N x H]) from the fused score/CE criterion.  Accumulates the option-LSTM gradients, leaves the
        gradient of the gathered table in self.dtab (the shared embedding gradient is added by backward_embed(), kept
        separate so the host can order it after the encoder's scatters) and returns [None, gradEncOut] like the
        reference's table of input gradients ({_, gradEncOut}, model.lua:335-337)."""
        otok, enc_out = inputs
        d_optH, d_enc = gradOutput
        ws, H, V, To, NO = self.ws, self.H, self.V, self.To, self.NO
        if self.uid is not None:       # the gradients of the copies of a distinct row add up
            d_u = ws.get('opt.d_optH_u', (NO, H))
            ops.zero(d_u)
            ops.embed_scatter_acc(d_u, self.uid, d_optH)
            d_optH = d_u
        dc = ws.get('opt.dc', (NO, H))
        # the token counting sort and the zero-fill of the table gradient depend on the inputs only:
        # they run on a side stream underneath the backward recurrence
        tokf = otok.view(-1)
        offset = ws.get('opt.sort_off', (V + 2,), torch.int32)
        work = ws.get('opt.sort_work', (2 * (V + 1),), torch.int32)
        perm = ws.get('opt.sort_perm', (To * NO,), torch.int32)
        dtab = ws.get('opt.dtable', (V + 1, 4 * H))
        with self.streams.fork('tab'):
            ops.token_sort(tokf, V + 1, offset, work, perm)
            ops.zero(dtab)
        t0 = ops.prof_begin('opt_lstm_bwd')
        ops.lstm_backward(self.Wh, self.gates, self.c, dc, To, NO, H, dh_last=d_optH, flags=self.flags)
        ops.prof_end('opt_lstm_bwd', t0, 1)
        da = self.gates.view(To * NO, 4 * H)
        # gradient of the gathered table (segmented row sums over the token-sorted rows: one HBM-bound pass over da)
        # and its two small consumers (bias, input weights) run on the side stream BESIDE the MFMA-bound dWh
        # contraction of the main stream; both only need the finished recurrence
        with self.streams.fork('tab'):
            ops.segment_rowsum_acc(da, tokf, perm, dtab)
            ops.colsum_acc(dtab, self.db, M=V + 1, N=4 * H)
            ops.gemm_tn_acc(self.emb, dtab, self.dWx, M=self.E, N=4 * H, K=V + 1)
            # dEmb += dTable * Wx^T with float atomics (accumulate = 2): the SHARED embedding gradient has concurrent
            # atomic writers (the encoder's scatters); off the main stream's critical path here
            ops.gemm_nt(dtab, self.Wx, self.demb, accumulate=2, M=V + 1, N=self.E, K=4 * H)
        if To > 1:
            t0 = ops.prof_begin('opt_lstm_dWh')
            ops.gemm_tn_acc(self.h.view(To * NO, H), da[NO:], self.dWh, M=H, N=4 * H, K=(To - 1) * NO,
                            flags=self.flags)
            ops.prof_end('opt_lstm_dWh', t0, 1)
        self.dtab = dtab
        return [None, d_enc]

    def backward_embed(self):
        """the table-gradient stream (incl. dEmb += dTable * Wx^T) joins the main stream"""
        self.streams.join('tab')


def model(params, enc, fp, ws, drop):
    return Decoder(params, enc, fp, ws, drop)


def forwardConnect(enc, dec, encOut, seqLen):      # disc.lua:35 -- no-op
    pass


def backwardConnect(enc, dec):                     # disc.lua:38 -- no-op
    pass
