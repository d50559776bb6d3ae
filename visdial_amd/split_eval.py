"""Split-level loops of the reference's Model (model.lua:109-246) shared by both hosts: Model:evaluate (validation loss /
perplexity), Model:retrieve (ground-truth ranks + R@k / MRR) and Model:predict (all 100 ranks per round) over the
sequential batches of `dataloader:getTestBatch`.  A host provides `params`, `_set_training(bool)`,
`forwardBackward(batch, onlyForward=True)` and `retrieveBatch(batch)` (ranks per `params['useGt']`)."""
import math

import numpy as np

from . import utils


class SplitEval(object):
    def evaluate(self, dataloader, dtype):
        """model.lua:109-139: validation loss / perplexity over a split (generative decoder: summed token NLL over
        the number of non-pad target tokens; for the discriminative decoder, whose batches carry no answer_out, the
        reference would fail -- here the mean cross-entropy over rounds is reported instead).  Returns (loss, ppl)."""
        self._set_training(False)
        n = dataloader.numThreads[dtype]
        cur, count, start = 0.0, 0.0, 1
        while start <= n:
            batch, nxt = dataloader.getTestBatch(start, self.params, dtype)
            if self.params['decoder'] == 'gen':
                count += float((batch['answer_out'] > 0).sum())
                cur += self.forwardBackward(batch, onlyForward=True)
            else:
                rounds = float(np.asarray(batch['answer_ind']).size)
                count += rounds
                cur += self.forwardBackward(batch, onlyForward=True) * rounds
            start = nxt
        cur /= max(count, 1.0)
        print('\n%s\tLoss: %f\t Perplexity: %f\n' % (dtype, cur, math.exp(cur)))
        self._set_training(True)
        return cur, math.exp(cur)

    def _rank_records(self, dataloader, dtype, ranks, last_round_only):
        """{image_id, round_id, ranks} records as model.lua:174-184 / :222-241 builds them: real image ids, only the
        rounds that exist (num_rounds), and for the test split of predict() the last round only."""
        ids = getattr(dataloader, 'unique_img_' + dtype, None)
        rounds = getattr(dataloader, dtype + '_num_rounds', None)
        n, R = ranks.shape[0], ranks.shape[1]
        out = []
        for i in range(n):
            iid = ids[i] if ids is not None and i < len(ids) else i + 1
            nr = int(rounds[i]) if rounds is not None else R
            if last_round_only:
                out.append({'image_id': iid, 'round_id': nr, 'ranks': ranks[i, nr - 1].tolist()})
            else:
                for j in range(nr):
                    r = ranks[i, j]
                    out.append({'image_id': iid, 'round_id': j + 1, 'ranks': r.tolist() if np.ndim(r) else float(r)})
        return out

    def retrieve(self, dataloader, dtype):
        """model.lua:142-189: ground-truth ranks + metrics.  Returns (metrics, records)."""
        self._set_training(False)
        self.params['useGt'] = True
        n = dataloader.numThreads[dtype]
        R = int(self.params['maxQuesCount'])
        O = int(self.params.get('numOptions', 100))
        ranks = np.full((n, R), O + 1.0)                               # model.lua:153-154
        start = 1
        while start <= n:
            batch, nxt = dataloader.getTestBatch(start, self.params, dtype)
            ranks[start - 1:nxt - 1] = np.asarray(self.retrieveBatch(batch)).reshape(-1, R)
            start = nxt
        print('\n%s - Retrieval:' % dtype)
        metrics = utils.processRanks(ranks)
        self._set_training(True)
        return metrics, self._rank_records(dataloader, dtype, ranks, False)

    def predict(self, dataloader, dtype):
        """model.lua:192-246: all 100 ranks per round (val: every existing round; test: the last round only)."""
        self._set_training(False)
        self.params['useGt'] = False
        n = dataloader.numThreads[dtype]
        R = int(self.params['maxQuesCount'])
        O = int(self.params.get('numOptions', 100))
        ranks = np.full((n, R, O), O + 1.0)
        start = 1
        while start <= n:
            batch, nxt = dataloader.getTestBatch(start, self.params, dtype)
            ranks[start - 1:nxt - 1] = np.asarray(self.retrieveBatch(batch)).reshape(-1, R, O)
            start = nxt
        self._set_training(True)
        return self._rank_records(dataloader, dtype, ranks, dtype == 'test')
