"""Synthetic VisDial-shaped batches with the exact tensor layout of the reference dataloader.

Mirrors the batch CONTRACT of dataloader.lua:324-339 (getTrainBatch), :378-432 (getIndexData)
and :463-475 (getIndexOption) -- keys, shapes, alignment and padding -- without reading
HDF5/JSON (no dataset offline):

  ques_fwd   int32 [B, R, Tq]   right-aligned, left-padded with 0, trimmed to the batch max
  hist       int32 [B, R, Th]   right-aligned (round 1 = caption, round r = QA of r-1)
  img_feat   fp32  [B, S, S, C] (att encoders, pool5, non-negative) or [B, F] (fc7, L2-normalised)
  options    int32 [B*R, O, To] left-aligned, trailing zeros, no <START>/<END> (disc)
  answer_ind int32 [B*R]        1-based index of the ground-truth option (prepro.py:169)
  answer_in / answer_out int32 [B, R, Ta]  <START>+tokens / tokens+<END>, left-aligned (gen)
"""
import numpy as np


def _right_align(rows, T):
    out = np.zeros((len(rows), T), np.int32)
    for i, r in enumerate(rows):
        if len(r):
            out[i, T - len(r):] = r
    return out


def _left_align(rows, T):
    out = np.zeros((len(rows), T), np.int32)
    for i, r in enumerate(rows):
        out[i, :len(r)] = r
    return out


class SyntheticDataloader(object):
    """Stand-in for `dataloader` (dataloader.lua): same attributes the model reads
    (vocabSize, maxQuesCount, maxQuesLen, maxAnsLen, numThreads) and getTrainBatch/getTestBatch."""

    def __init__(self, opt, seed=1234, num_threads=None):
        self.opt = opt
        self.vocabSize = int(opt.get('vocabSize', 11322))   # incl. <START>, <END> (dataloader.lua:17-22)
        self.maxQuesCount = int(opt.get('maxQuesCount', 10))
        self.maxQuesLen = int(opt.get('maxQuesLen', 20))
        self.maxAnsLen = int(opt.get('maxAnsLen', 20))
        self.maxHistLen = int(opt.get('maxHistoryLenPerRound', 40))
        self.numOptions = int(opt.get('numOptions', 100))
        self.rng = np.random.RandomState(seed)
        n = int(num_threads or opt.get('numTrainThreads', 1000))
        self.numThreads = {'train': n, 'val': n, 'test': n}
        self.numTrainThreads = n
        self.numValThreads = n
        self.numTestThreads = n
        self.startToken = self.vocabSize - 1   # <START> = V-1, <END> = V (ids as in dataloader.lua:17-22)
        self.endToken = self.vocabSize

    def _tokens(self, n):
        return self.rng.randint(1, self.vocabSize - 1, size=n).astype(np.int32)

    def getTrainBatch(self, params, batch_size=None, full_length=True):
        B = int(batch_size or params['batchSize'])
        R, V = self.maxQuesCount, self.vocabSize
        rng = self.rng
        N = B * R
        batch = {}
        qlen = rng.randint(3, self.maxQuesLen + 1, size=N)
        if full_length:
            qlen[rng.randint(N)] = self.maxQuesLen
        Tq = int(qlen.max())
        batch['ques_fwd'] = _right_align([self._tokens(l) for l in qlen], Tq).reshape(B, R, Tq)
        enc = params['encoder']
        if 'hist' in enc:
            if enc.startswith('lf'):   # concatenated history (opts.lua:59, dataloader.lua:217-221,243-255)
                maxlen = int(params.get('maxHistoryLen', 60)) * 5
                hl = np.minimum(np.cumsum(rng.randint(8, 31, size=(B, R)), axis=1), maxlen).reshape(-1)
            else:
                hl = rng.randint(4, self.maxHistLen + 1, size=N)
                hl[::R] = rng.randint(5, self.maxHistLen + 1, size=B)     # captions
                if full_length:
                    hl[rng.randint(N)] = self.maxHistLen
            Th = int(hl.max())
            batch['hist'] = _right_align([self._tokens(l) for l in hl], Th).reshape(B, R, Th)
        if 'im' in enc:
            if 'att' in enc:
                S, C = int(params['imgSpatialSize']), int(params['imgFeatureSize'])
                batch['img_feat'] = np.abs(rng.randn(B, S, S, C)).astype(np.float32)
            else:
                f = rng.randn(B, int(params['imgFeatureSize'])).astype(np.float32)
                if int(params.get('imgNorm', 1)) == 1:
                    f /= np.linalg.norm(f, axis=1, keepdims=True)
                batch['img_feat'] = f
        if params['decoder'] == 'disc':
            O, To = self.numOptions, self.maxAnsLen
            ol = rng.randint(1, To + 1, size=N * O)
            opts = _left_align([self._tokens(l) for l in ol], To).reshape(N, O, To)
            batch['options'] = opts
            batch['answer_ind'] = rng.randint(1, O + 1, size=N).astype(np.int32)
        else:
            al = rng.randint(1, self.maxAnsLen + 1, size=N)
            Ta = int(al.max()) + 1
            toks = [self._tokens(l) for l in al]
            batch['answer_in'] = _left_align([np.concatenate([[self.startToken], t]) for t in toks], Ta).reshape(B, R, Ta)
            batch['answer_out'] = _left_align([np.concatenate([t, [self.endToken]]) for t in toks], Ta).reshape(B, R, Ta)
        return batch

    def getTestBatch(self, start_id, params, dtype='val'):
        """dataloader.lua:342-375: sequential batches; returns (batch, nextStartId).  For the gen decoder the
        100 candidates arrive as option_in = <START>+tokens / option_out = tokens+<END> [B, R, O, To+1]
        (processOptions, dataloader.lua:281-318; getIndexOption :437-462) plus answer_ind."""
        B = int(params['batchSize'])
        n = self.numThreads[dtype]
        nb = min(B, n - start_id + 1)
        batch = self.getTrainBatch(params, batch_size=nb, full_length=False)
        if params['decoder'] == 'gen':
            self.add_gen_options(batch, nb)
        return batch, start_id + nb

    def add_gen_options(self, batch, B):
        R, O, To = self.maxQuesCount, self.numOptions, self.maxAnsLen
        N = B * R
        ol = self.rng.randint(1, To + 1, size=N * O)
        toks = [self._tokens(l) for l in ol]
        T = int(ol.max()) + 1
        batch['option_in'] = _left_align([np.concatenate([[self.startToken], t]) for t in toks], T).reshape(B, R, O, T)
        batch['option_out'] = _left_align([np.concatenate([t, [self.endToken]]) for t in toks], T).reshape(B, R, O, T)
        batch['answer_ind'] = self.rng.randint(1, O + 1, size=N).astype(np.int32)
        return batch


def dropout_mask_shapes(params, batch):
    """Shapes of the nn.Dropout keep-masks each encoder file instantiates (name -> shape), for hosts that
    pin the noise (parity runs).  Names are the call sites: q_emb / h_emb (embedding dropout of the
    nngraph encoders), hatt, img_tr, iqc, u (mn-att:24-25,64,74,92,106), fuse (lf-*), img (hrea:47)."""
    enc = params['encoder']
    B, R, Tq = batch['ques_fwd'].shape
    N, H, E = B * R, params['rnnHiddenSize'], params['embedSize']
    shp = {}
    if enc.startswith('lf-ques'):
        D = H * (2 if 'hist' in enc else 1) + (params['imgFeatureSize'] if 'im' in enc else 0)
        if params.get('dropout', 0.5) > 0:
            shp['fuse'] = (N, D)
    elif enc.startswith('hrea'):
        shp['img'] = (N, params['imgFeatureSize'])
    elif enc.startswith('mn') or enc.startswith('lf-att'):
        Th = batch['hist'].shape[2]
        shp.update(q_emb=(Tq, N, E), h_emb=(Th, N, E))
        if enc.startswith('mn'):
            shp['hatt'] = (N, H)
        if 'att' in enc:
            S2, K = params['imgSpatialSize'] ** 2, params.get('commonEmbeddingSize', 512)
            shp.update(img_tr=(N, S2, H), iqc=(N, S2, K), u=(N, H))
    return shp
