"""Split-level loops of the reference's Model (model.lua:109-246, 432-613) shared by both hosts: Model:evaluate (validation loss /
perplexity), Model:retrieve (ground-truth ranks + R@k / MRR) and Model:predict (all 100 ranks per round) over the
sequential batches of `dataloader:getTestBatch`.  A host provides `params`, `_set_training(bool)`,
`forwardBackward(batch, onlyForward=True)` and `retrieveBatch(batch)` (ranks per `params['useGt']`); for
Model:generateAnswers the four device steps `_gen_encode(batch)`, `_gen_begin(rounds)`, `_gen_step(tokens) -> logp`,
`_gen_select(src, n_keep)` (= vd_model_encode / decode_begin / decode_step / decode_select of the model-level ABI)."""
import math

import numpy as np

from . import utils


class SplitEval(object):
    def evaluate(self, dataloader, dtype):
        """model.lua:109-139: validation loss / perplexity over a split: the sum over batches of `forwardBackward(batch, true)` (gen:
        summed token NLL; disc: the batch's MEAN cross-entropy) divided by the number of non-pad target tokens -- for both decoders,
        as the reference does (dataloader.lua:398-421 puts answer_out into every batch).  Synthetic disc batches without answer_out:
        the mean cross-entropy over rounds.  Returns (loss, ppl)."""
        self._set_training(False)
        n = dataloader.numThreads[dtype]
        cur, count, start = 0.0, 0.0, 1
        while start <= n:
            batch, nxt = dataloader.getTestBatch(start, self.params, dtype)
            if 'answer_out' in batch:
                count += float((np.asarray(batch['answer_out']) > 0).sum())
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

    # ------------------------------------------------------------------ generation (model.lua:432-613)
    def generateAnswers(self, dataloader, dtype, params=None):
        """Beam search (default) or temperature sampling with the generative decoder, one dialog at a time,
        exactly as the reference drives it from the host: the decoder step (embedding, LSTM stack, vocabulary
        projection, log-softmax) runs on the device for all hypotheses at once, candidate bookkeeping is host
        control flow.  Returns [{image_id, dialog: [{question, answer}...]}]."""
        if self.params['decoder'] == 'disc':
            raise SystemExit('Sampling/beam search only for generative model')
        params = params or {}
        sampleWords = bool(params.get('sampleWords', 0) == 1)
        temperature = float(params.get('temperature', 1.0))
        beamSize, beamLen = int(params.get('beamSize', 5)), int(params.get('beamLen', 20))
        startToken, endToken = dataloader.word2ind['<START>'], dataloader.word2ind['<END>']
        numThreads = int(params.get('maxThreads') or dataloader.numThreads[dtype])
        rng = np.random.RandomState(int(params.get('seed', 1234)))
        ind2word = dataloader.ind2word
        answerTable = []
        self._set_training(False)
        for convId in range(1, numThreads + 1):
            batch = dataloader.getIndexData(np.array([convId]), self.params, dtype)
            R = batch['ques_fwd'].shape[1]
            self._gen_encode(batch)                                               # forwardBackward(batch, true, true)
            threadAnswers = []
            if not sampleWords:
                for it in range(R):
                    beams = np.zeros((beamLen, beamSize), np.int64)
                    self._gen_begin(np.full(beamSize, it, np.int32))              # hiddenBeams, model.lua:478-503
                    beams[0] = startToken
                    scores = np.zeros(beamSize)
                    finish = []
                    for step in range(1, beamLen):
                        exploreSize = 1 if step == 1 else beamSize                    # all beams are <START> at first
                        logp = self._gen_step(beams[step - 1])
                        cands = []
                        for wordId in range(exploreSize):
                            top = np.argsort(-logp[wordId], kind='stable')[:beamSize]  # torch.topk(..., true)
                            for cid in top:
                                cb = beams[:, wordId].copy()
                                cb[step] = cid + 1                                     # vocabulary ids are 1-based
                                sc = scores[wordId] + float(logp[wordId, cid])
                                if cid + 1 == endToken:
                                    finish.append(dict(beam=cb, length=step + 1, score=sc))
                                else:
                                    cands.append(dict(score=sc, beam=cb, src=wordId))
                        cands.sort(key=lambda a: -a['score'])                         # (stable; Lua's table.sort is not)
                        keep = cands[:beamSize]
                        if keep:                                                      # untouched slots keep their old state
                            self._gen_select(np.array([c['src'] for c in keep], np.int32), len(keep))
                        for i, c in enumerate(keep):
                            beams[:, i] = c['beam']
                            scores[i] = c['score']
                    finish.sort(key=lambda a: -a['score'])
                    best = finish[0]['beam'] if finish else beams[:, 0]               # (the reference errors if none ended)
                    threadAnswers.append({'question': utils.idToWords(batch['ques_fwd'][0, it], ind2word),
                                          'answer': utils.idToWords(best, ind2word)})
            else:
                numQues = R
                self._gen_begin(np.arange(R, dtype=np.int32))
                answerIn = np.full(numQues, startToken, np.int64)
                answer = [answerIn[:, None].copy()]
                for timeStep in range(beamLen):
                    logp = self._gen_step(answerIn)
                    self._gen_select(np.arange(numQues, dtype=np.int32), numQues)
                    pr = np.exp(logp.astype(np.float64) / temperature)
                    pr /= pr.sum(1, keepdims=True)
                    nxt = np.array([rng.choice(pr.shape[1], p=pr[i]) + 1 for i in range(numQues)], np.int64)
                    answer.append(nxt[:, None])
                    answerIn = nxt
                answer = np.concatenate(answer, 1)
                for it in range(R):
                    threadAnswers.append({'question': utils.idToWords(batch['ques_fwd'][0, it], ind2word),
                                          'answer': utils.idToWords(answer[it], ind2word)})
            img_ids = getattr(dataloader, 'unique_img_' + dtype, None)
            answerTable.append({'image_id': img_ids[convId - 1] if img_ids else convId, 'dialog': threadAnswers})
        self._set_training(True)
        return answerTable
