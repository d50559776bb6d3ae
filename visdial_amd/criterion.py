"""Criteria -- counterparts of the reference's `self.criterion` objects (model.lua:32-38): same call surface
(`criterion:forward(decOut, target)` -> loss, `criterion:backward(decOut, target)` -> gradient w.r.t. decOut).

CrossEntropyCriterion (disc, model.lua:37-38,330,334): the decoder output handed to it is the *unreduced* pair
(option states, encoder output) -- decoders/disc.lua's final nn.MM is folded into the criterion kernel, so that scores,
softmax, NLL and both gradients of the bilinear product are ONE pass over optH (41 MB) instead of four.  `forward`
launches that kernel and returns a deferred loss (call it to wait for the device); `backward` hands out the gradients
the same launch produced: (d optH [N*O x H], d encOut [N x H]) -- what nn.MM:backward would return for its two inputs.
"""
import numpy as np
import torch

from . import ops


class DiscDecoderOutput(object):
    """decOut of the discriminative decoder: scores [N x O] (filled by the criterion / `materialize`), plus the MM
    operands the fused criterion consumes"""

    def __init__(self, optH, encOut, scores, N, O, H):
        self.optH, self.encOut, self.scores, self.N, self.O, self.H = optH, encOut, scores, N, O, H

    def materialize(self):
        """scores without a criterion (retrieval path, model.lua:415-417)"""
        ops.score_ce(self.optH, self.encOut, self.scores, self.N, self.O, self.H)
        return self.scores


class CrossEntropyCriterion(object):
    """nn.CrossEntropyCriterion (sizeAverage = true: mean over the N rounds; targets 0-based int32 on device)."""

    def __init__(self, ws):
        self.ws = ws
        self._host = None
        self.gradInput = None

    def forward(self, decOut, target, needGrad=True):
        N, O, H = decOut.N, decOut.O, decOut.H
        loss_rows = self.ws.get('crit.loss_rows', (N,))
        if needGrad:
            d_optH = self.ws.get('crit.d_optH', (N * O, H))
            d_enc = self.ws.get('crit.d_enc', (N, H))
            ops.score_ce(decOut.optH, decOut.encOut, decOut.scores, N, O, H, gt=target, loss_rows=loss_rows,
                         dOptH=d_optH, dEnc=d_enc, gscale=1.0 / N)
            self.gradInput = (d_optH, d_enc)
        else:
            ops.score_ce(decOut.optH, decOut.encOut, decOut.scores, N, O, H, gt=target, loss_rows=loss_rows)
            self.gradInput = None
        self.done = torch.cuda.Event()
        self.done.record()
        self._rows, self._n = loss_rows, N
        return self

    def loss_handle(self):
        """enqueue the D2H copy of the per-round losses (pinned buffer); returns a callable that waits and reduces"""
        N = self._n
        if self._host is None or self._host.numel() < N:
            self._host = torch.empty(N, dtype=torch.float32).pin_memory()
        host = self._host[:N]
        host.copy_(self._rows, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()

        def pending():
            ev.synchronize()
            return float(host.numpy().astype(np.float64).mean())
        return pending

    def backward(self, decOut, target):
        assert self.gradInput is not None, "criterion:forward(..., needGrad=True) first"
        return self.gradInput
