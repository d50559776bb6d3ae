"""Memory-network encoder with fc7 image: counterpart of encoders/mn-ques-im-hist.lua.
query = Tanh(Linear(H+F, H)(JoinTable{q3, img})) (:47-48), then the memory block of mn-ques-hist.
Inputs: ques [Tq x N], img [B x F], hist [Th x N], mask [N x R]."""
from .. import ops
from ._blocks import TextBranches, MemoryBlock, CatLinear, round_index


def declare(params, spec):
    H, F = params['rnnHiddenSize'], params['imgFeatureSize']
    TextBranches.declare(params, spec)
    spec.linear('qi', H + F, H)
    MemoryBlock.declare(params, spec)


class Encoder(object):
    def __init__(self, params, fp, ws, drop, streams=None):
        self.ws, self.H, self.F, self.R = ws, params['rnnHiddenSize'], params['imgFeatureSize'], params['maxQuesCount']
        self.wordEmbed = ('embed', fp.w['embed'], fp.g['embed'])
        self.text = TextBranches(params, fp, ws, drop)
        self.qi = CatLinear(fp, 'qi', [self.H, self.F], self.H, ws)
        self.memory = MemoryBlock(params, fp, ws, drop)
        self._idx = {}

    def output_buffer(self, inputs):
        return self.ws.get('mn2.y', (inputs[0].shape[1], self.H))

    def forward(self, inputs):
        ques, img, hist, mask = inputs
        N = ques.shape[1]
        q3, h3 = self.text.forward(ques, hist)
        img_rep = ops.embed_gather(img, round_index(N, self.R, ques.device, self._idx), self.ws.get('img.rep', (N, self.F)))
        qi_proj = self.qi.forward([q3, img_rep], N)
        self.output = self.memory.forward(qi_proj, h3, mask)
        return self.output

    def backward(self, inputs, grad_output):
        ques, img, hist, mask = inputs
        dquery, dh3 = self.memory.backward(grad_output)
        dq3 = self.qi.backward(dquery, need=[True, False])[0]
        self.text.backward(ques, hist, dq3, dh3)


def model(params, fp, ws, drop, streams=None):
    return Encoder(params, fp, ws, drop, streams)
