"""Memory-network + image-attention encoder: counterpart of encoders/mn-att-ques-im-hist.lua:5-115.

Inputs (order of the reference's input table, model.lua:255-294):
  ques [Tq x N] int32 time-major, img [B*S2 x C] fp32 (one map per IMAGE -- the reference's 10x
  repeatTensor copy, model.lua:262-265, is folded into the attention kernels' loaders), hist [Th x N],
  mask [N x R] uint8 (1 = hidden).  Output encOut [N x H].
Like the reference this encoder hard-codes two LSTM layers per branch and dropout 0.5 (mn-att:24-41)."""
from ..nn import StreamPool
from ._blocks import TextBranches, MemoryBlock, SANBlock


def declare(params, spec):
    TextBranches.declare(params, spec)       # mn-att:27-41
    MemoryBlock.declare(params, spec)        # mn-att:64-65
    SANBlock.declare(params, spec)           # mn-att:77,84,88,93,106


class Encoder(object):
    def __init__(self, params, fp, ws, drop, streams=None):
        self.ws, self.H = ws, params['rnnHiddenSize']
        self.streams = streams if streams is not None else StreamPool(None, enabled=False)
        self.wordEmbed = ('embed', fp.w['embed'], fp.g['embed'])      # enc.wordEmbed (mn-att:21,112)
        self.text = TextBranches(params, fp, ws, drop)
        self.memory = MemoryBlock(params, fp, ws, drop)
        self.san = SANBlock(params, fp, ws, drop, self.streams)

    def output_buffer(self, inputs):
        """the tensor forward() will return (lets the host enqueue consumers' launches first)"""
        return self.ws.get('out.y', (inputs[0].shape[1], self.H))

    def forward(self, inputs):
        ques, img, hist, mask = inputs
        N = ques.shape[1]
        self.san.prefetch(img, N)
        q3, h3 = self.text.forward(ques, hist)
        qh2 = self.memory.forward(q3, h3, mask)                       # mn-att:48-65
        self.output = self.san.forward(qh2)                           # mn-att:68-106
        return self.output

    def backward(self, inputs, grad_output):
        ques, img, hist, mask = inputs
        dqh2 = self.san.backward(grad_output)
        dq3, dh3 = self.memory.backward(dqh2)
        self.text.backward(ques, hist, dq3, dh3)


def model(params, fp, ws, drop, streams=None):
    return Encoder(params, fp, ws, drop, streams)
