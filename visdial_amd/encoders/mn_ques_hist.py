"""Memory-network encoder (question + history): counterpart of encoders/mn-ques-hist.lua.
Inputs: ques [Tq x N], hist [Th x N], mask [N x R] (model.lua:255-294 with useIm = false)."""
from ..nn import StreamPool
from ._blocks import TextBranches, MemoryBlock


def declare(params, spec):
    TextBranches.declare(params, spec)
    MemoryBlock.declare(params, spec)


class Encoder(object):
    def __init__(self, params, fp, ws, drop, streams=None):
        self.ws, self.H = ws, params['rnnHiddenSize']
        self.wordEmbed = ('embed', fp.w['embed'], fp.g['embed'])
        self.text = TextBranches(params, fp, ws, drop)
        self.memory = MemoryBlock(params, fp, ws, drop)

    def output_buffer(self, inputs):
        return self.ws.get('mn2.y', (inputs[0].shape[1], self.H))

    def forward(self, inputs):
        ques, hist, mask = inputs
        q3, h3 = self.text.forward(ques, hist)
        self.output = self.memory.forward(q3, h3, mask)
        return self.output

    def backward(self, inputs, grad_output):
        ques, hist, mask = inputs
        dq3, dh3 = self.memory.backward(grad_output)
        self.text.backward(ques, hist, dq3, dh3)


def model(params, fp, ws, drop, streams=None):
    return Encoder(params, fp, ws, drop, streams)
