"""Late-fusion question-only encoder: counterpart of encoders/lf-ques.lua:3-36.
embed -> numLayers x SeqLSTM(maskZero) -> last step -> Dropout(p) -> Linear(H,H) -> Tanh.
Exposes wordEmbed and rnnLayers (read by decoders/gen.lua:30-60)."""
from .. import ops
from ..nn import SeqLSTM, Linear, dropout_forward, dropout_backward, lstm_stack_forward, lstm_stack_backward


def declare(params, spec):
    E, H = params['embedSize'], params['rnnHiddenSize']
    for l in range(params['numLayers']):
        spec.lstm('ques%d' % (l + 1), E if l == 0 else H, H)        # lf-ques.lua:17-24
    spec.linear('fuse', H, H)                                       # lf-ques.lua:32


class Encoder(object):
    def __init__(self, params, fp, ws, drop, streams=None):
        self.p, self.fp, self.ws, self.drop = params, fp, ws, drop
        E, H = params['embedSize'], params['rnnHiddenSize']
        self.E, self.H = E, H
        self.pdrop = float(params.get('dropout', 0.5))
        self.wordEmbed = ('embed', fp.w['embed'], fp.g['embed'])
        self.rnnLayers = [SeqLSTM(fp, 'ques%d' % (l + 1), E if l == 0 else H, H, ws) for l in range(params['numLayers'])]
        self.fuse = Linear(fp, 'fuse', H, H, ws)

    def output_buffer(self, inputs):
        return self.ws.get('fuse.y', (inputs[0].shape[1], self.H))

    def forward(self, inputs):
        ques = inputs[0]
        T, N = ques.shape
        x = self.ws.get('q.x', (T * N, self.E))
        ops.embed_gather(self.fp.w['embed'], ques, x)
        h = lstm_stack_forward(self.rnnLayers, x, T, N, ques)
        self.scale = 1.0 / (1.0 - self.pdrop) if self.pdrop > 0 else 1.0
        self.m = self.drop.mask('fuse', N * self.H, self.pdrop)
        hd = dropout_forward(self.ws, 'fuse.in', h[T - 1], self.m, self.scale)
        self.output = self.fuse.forward(hd, N, tanh=True)
        return self.output

    def backward(self, inputs, grad_output):
        ques = inputs[0]
        dhd = self.fuse.backward(grad_output)
        dh = dropout_backward(self.ws, 'fuse.din', dhd, self.m, self.scale)
        dx = lstm_stack_backward(self.rnnLayers, dh_last_top=dh)[0]
        ops.embed_scatter_acc(self.fp.g['embed'], ques, dx)


def model(params, fp, ws, drop, streams=None):
    return Encoder(params, fp, ws, drop, streams)
