"""Late fusion with image attention: counterpart of encoders/lf-att-ques-im-hist.lua.
qh = Tanh(Linear(2H, H)(JoinTable{q3, h3})) (:43), then the same 1-hop SAN attention as mn-att (:45-86).
Inputs: ques [Tq x N], img [B*S2 x C], hist [Th x N].  Exposes wordEmbed only (no rnnLayers), like the reference."""
from ..nn import StreamPool
from ._blocks import TextBranches, SANBlock, CatLinear


def declare(params, spec):
    H = params['rnnHiddenSize']
    TextBranches.declare(params, spec)
    spec.linear('qh', 2 * H, H)
    SANBlock.declare(params, spec)


class Encoder(object):
    def __init__(self, params, fp, ws, drop, streams=None):
        self.ws, self.H = ws, params['rnnHiddenSize']
        self.streams = streams if streams is not None else StreamPool(None, enabled=False)
        self.wordEmbed = ('embed', fp.w['embed'], fp.g['embed'])
        self.text = TextBranches(params, fp, ws, drop)
        self.qh = CatLinear(fp, 'qh', [self.H, self.H], self.H, ws)
        self.san = SANBlock(params, fp, ws, drop, self.streams)

    def output_buffer(self, inputs):
        return self.ws.get('out.y', (inputs[0].shape[1], self.H))

    def forward(self, inputs):
        ques, img, hist = inputs
        N = ques.shape[1]
        self.san.prefetch(img, N)
        q3, h3 = self.text.forward(ques, hist)
        u = self.qh.forward([q3, h3], N)
        self.output = self.san.forward(u)
        return self.output

    def backward(self, inputs, grad_output):
        ques, img, hist = inputs
        du = self.san.backward(grad_output)
        dq3, dh3 = self.qh.backward(du)
        self.text.backward(ques, hist, dq3, dh3)


def model(params, fp, ws, drop, streams=None):
    return Encoder(params, fp, ws, drop, streams)
