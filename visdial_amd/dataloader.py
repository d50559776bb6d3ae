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
import re

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


def _aligned_random(rng, lens, T, lo, hi, right):
    """[len(lens) x T] int32 token matrix, row i holding lens[i] ids from [lo, hi) right- or left-aligned, 0 elsewhere
    (vectorised equivalent of _right_align / _left_align over per-row random draws)"""
    lens = np.asarray(lens)
    tok = rng.randint(lo, hi, size=(len(lens), T)).astype(np.int32)
    pos = np.arange(T)[None, :]
    keep = (pos >= T - lens[:, None]) if right else (pos < lens[:, None])
    tok *= keep
    return tok


class SyntheticDataloader(object):
    """Stand-in for `dataloader` (dataloader.lua): same attributes the model reads
    (vocabSize, maxQuesCount, maxQuesLen, maxAnsLen, numThreads) and getTrainBatch/getTestBatch.
    fast=True draws every token matrix in one vectorised call (same layout contract and distributions, a
    different random stream): a fresh full-size batch costs a few ms of host time instead of ~0.2 s, which is
    what lets bench.py draw a new batch every step like the reference's trainIteration does (model.lua:71)."""

    def __init__(self, opt, seed=1234, num_threads=None, fast=False):
        self.opt = opt
        self.fast = bool(fast)
        self.gen = np.random.default_rng(seed) if fast else None
        self._img_pool = None
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
        for split in ('train', 'val', 'test'):  # fields Model:retrieve / predict read (model.lua:174-184, 222-241)
            setattr(self, 'unique_img_' + split, list(range(1, n + 1)))
            setattr(self, split + '_num_rounds', np.full(n, self.maxQuesCount, np.int64))

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
        if self.fast:
            batch['ques_fwd'] = _aligned_random(rng, qlen, Tq, 1, V - 1, True).reshape(B, R, Tq)
        else:
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
            if self.fast:
                batch['hist'] = _aligned_random(rng, hl, Th, 1, V - 1, True).reshape(B, R, Th)
            else:
                batch['hist'] = _right_align([self._tokens(l) for l in hl], Th).reshape(B, R, Th)
        if 'im' in enc:
            if 'att' in enc:
                S, C = int(params['imgSpatialSize']), int(params['imgFeatureSize'])
                if self.fast:
                    # like the reference (dataloader.lua:378-390: `img_fv:index(1, inds)`): B rows of a resident
                    # image-feature tensor picked by random thread ids, not freshly drawn noise
                    if self._img_pool is None or self._img_pool.shape[1:] != (S, S, C):
                        self._img_pool = np.abs(self.gen.standard_normal((max(64, 2 * B), S, S, C), dtype=np.float32))
                    batch['img_feat'] = self._img_pool[self.gen.integers(0, len(self._img_pool), size=B)]
                else:
                    batch['img_feat'] = np.abs(rng.randn(B, S, S, C)).astype(np.float32)
            else:
                f = rng.randn(B, int(params['imgFeatureSize'])).astype(np.float32)
                if int(params.get('imgNorm', 1)) == 1:
                    f /= np.linalg.norm(f, axis=1, keepdims=True)
                batch['img_feat'] = f
        if params['decoder'] == 'disc':
            O, To = self.numOptions, self.maxAnsLen
            ol = rng.randint(1, To + 1, size=N * O)
            if self.fast:
                opts = _aligned_random(rng, ol, To, 1, V - 1, False).reshape(N, O, To)
            else:
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
            hops = 1 if enc.startswith('lf-att') else int(params.get('numAttentionLayers', 1) or 1)     # (lf-att-ques-im-hist.lua:49: always 1)
            for i in range(2, hops + 1):     # one Dropout per attention hop
                shp['iqc%d' % i] = (N, S2, K)
    return shp


# =====================================================================================================
# Real-data loader: counterpart of the reference's dataloader.lua.
# =====================================================================================================
def rightAlign(sequences, lengths):
    """utils.lua:6-45.  3-D case reproduces the reference's quirk: the inner loop BREAKS at the first
    zero-length sequence of an image, leaving that image's remaining rounds all-zero."""
    seq = np.asarray(sequences)
    out = np.zeros_like(seq)
    M = seq.shape[-1]
    if seq.ndim == 3:
        L = np.asarray(lengths).astype(np.int64)
        alive = np.cumprod(L != 0, axis=1).astype(bool)              # False from the first zero length on
        pos = np.arange(M)[None, None, :]
        src = pos - (M - L[:, :, None])                              # source column for every target column
        ok = (src >= 0) & alive[:, :, None]
        out[ok] = np.take_along_axis(seq, np.clip(src, 0, M - 1), axis=2)[ok]
    else:
        L = np.asarray(lengths).astype(np.int64)
        pos = np.arange(M)[None, :]
        src = pos - (M - L[:, None])
        ok = (src >= 0) & (L[:, None] > 0)
        out[ok] = np.take_along_axis(seq, np.clip(src, 0, M - 1), axis=1)[ok]
    return out


def _open_arrays(path):
    """The reference's HDF5 inputs (dataloader.lua:36-140 reads them with torch-hdf5).  Order of preference:
    an explicit .npz; h5py if this interpreter has it; visdial_amd.h5lite (ctypes over the libhdf5 C library);
    an .npz twin with the same dataset names (scripts/h5_to_npz.py) when no HDF5 library exists at all."""
    if path.endswith('.npz'):
        return np.load(path)
    try:
        import h5py
        return h5py.File(path, 'r')
    except ImportError:
        pass
    import os
    from . import h5lite
    if os.path.exists(path) and h5lite.available():
        return h5lite.File(path)
    alt = path[:-3] + '.npz' if path.endswith('.h5') else path + '.npz'
    if os.path.exists(alt):
        return np.load(alt)
    raise RuntimeError("cannot read %s: no h5py, no libhdf5 (set VD_HDF5_LIB) and no %s twin "
                       "(convert with scripts/h5_to_npz.py)" % (path, alt))


class Dataloader(object):
    """Same public surface as the reference `dataloader` table: initialize(opt, subsets),
    getTrainBatch(params[, batchSize]), getTestBatch(startId, params, dtype), getIndexData,
    getIndexOption, and the size fields train.lua copies into the model params (train.lua:55-59).
    Token matrices stay 1-based vocabulary ids with 0 = pad, exactly as prepro.py writes them."""

    def __init__(self, seed=1234):
        self.rng = np.random.RandomState(seed)

    # -------------------------------------------------------------------------------- loading
    def initialize(self, opt, subsets):
        import json
        info = json.load(open(opt['inputJson']))
        ques = _open_arrays(opt['inputQues'])
        img = _open_arrays(opt['inputImg']) if opt.get('useIm') else None
        return self.from_arrays(info, ques, img, opt, subsets)

    def from_arrays(self, info, ques, img, opt, subsets):
        """dataloader.lua:10-140"""
        for k, v in info.items():
            setattr(self, k, v)
        self.word2ind = dict(info['word2ind'])
        count = len(self.word2ind)
        self.word2ind['<START>'] = count + 1                       # :17-22
        self.word2ind['<END>'] = count + 2
        self.vocabSize = count + 2
        self.ind2word = {i: w for w, i in self.word2ind.items()}
        self.numThreads = {}
        self.data = {}
        self.useHistory, self.concatHistory = bool(opt.get('useHistory')), bool(opt.get('concatHistory'))
        self.useIm = bool(opt.get('useIm'))
        self.maxHistoryLen = int(opt.get('maxHistoryLen') or 60)
        A = lambda name: np.asarray(ques[name])
        for dtype in subsets:
            d = {}
            # image ids to numbers (dataloader.lua:42-45: tonumber(string.match(v, '000%d+')))
            ids = getattr(self, 'unique_img_' + dtype, None)
            if ids is not None:
                conv = []
                for v in ids:
                    m = re.search(r'000\d+', str(v))
                    conv.append(int(m.group()) if m else v)
                setattr(self, 'unique_img_' + dtype, conv)
            d['ques'], d['ques_len'] = A('ques_' + dtype).astype(np.int64), A('ques_length_' + dtype).astype(np.int64)
            d['ans'], d['ans_len'] = A('ans_' + dtype).astype(np.int64), A('ans_length_' + dtype).astype(np.int64)
            if dtype != 'test':
                d['ans_ind'] = A('ans_index_' + dtype).astype(np.int64)
            if self.useIm:
                f = np.asarray(img['images_' + dtype], dtype=np.float32)
                if int(opt.get('imgNorm', 1)) == 1:                 # :64-68 (L2 norm over dim 2)
                    f = f / np.sqrt((f * f).sum(1, keepdims=True))
                if 'att' in opt['encoder']:                        # :69-72 NCHW -> NHWC
                    f = np.ascontiguousarray(f.transpose(0, 2, 3, 1))
                d['img_fv'] = f
                d['img_pos'] = A('img_pos_' + dtype).astype(np.int64) + 1     # :76-77 (1-indexed like Lua)
            n = d['ques'].shape[0]
            self.numThreads[dtype] = n
            setattr(self, 'num%sThreads' % dtype.capitalize(), n)
            d['opt'] = A('opt_' + dtype).astype(np.int64)
            d['opt_len'] = A('opt_length_' + dtype).astype(np.int64)
            d['opt_list'] = A('opt_list_' + dtype).astype(np.int64)
            self.numOptions = d['opt'].shape[2]
            d['num_rounds'] = A('num_rounds_' + dtype).astype(np.int64)
            setattr(self, dtype + '_num_rounds', d['num_rounds'])            # dataloader.lua:110
            self.maxQuesCount, self.maxQuesLen = d['ques'].shape[1], d['ques'].shape[2]
            self.maxAnsLen = d['ans'].shape[2]
            if self.useHistory:
                d['cap'], d['cap_len'] = A('cap_' + dtype).astype(np.int64), A('cap_length_' + dtype).astype(np.int64)
            self.data[dtype] = d
        for dtype in subsets:
            self.prepareDataset(dtype)
        return self

    # -------------------------------------------------------------------------------- preprocessing
    def prepareDataset(self, dtype):
        """dataloader.lua:143-156"""
        d = self.data[dtype]
        d['ques_fwd'] = rightAlign(d['ques'], d['ques_len'])
        if self.useHistory:
            self.processHistory(dtype)
        self.processOptions(dtype)
        self.processAnswers(dtype)

    def processAnswers(self, dtype):
        """dataloader.lua:159-200: <START>+answer / answer+<END>; <END> is written even for empty answers"""
        d = self.data[dtype]
        ans, L = d['ans'], d['ans_len']
        n, R, M = ans.shape
        din = np.zeros((n, R, M + 1), np.int64)
        dout = np.zeros((n, R, M + 1), np.int64)
        din[:, :, 0] = self.word2ind['<START>']
        keep = np.arange(M)[None, None, :] < L[:, :, None]           # the first `length` tokens
        din[:, :, 1:][keep] = ans[keep]
        dout[:, :, :M][keep] = ans[keep]
        ii, rr = np.meshgrid(np.arange(n), np.arange(R), indexing='ij')
        dout[ii, rr, L] = self.word2ind['<END>']
        d['ans_len'] = L + 1
        d['ans_in'], d['ans_out'] = din, dout

    def processOptions(self, dtype):
        """dataloader.lua:281-321: the same wrapping for the option list (used by the gen decoder only)"""
        d = self.data[dtype]
        ol, L = d['opt_list'], d['opt_len']
        n, M0 = ol.shape
        M = self.maxAnsLen
        din = np.zeros((n, M + 1), np.int64)
        dout = np.zeros((n, M + 1), np.int64)
        din[:, 0] = self.word2ind['<START>']
        keep = (np.arange(M)[None, :] < L[:, None])
        src = ol[:, :M] if M0 >= M else np.pad(ol, ((0, 0), (0, M - M0)))
        din[:, 1:][keep] = src[keep]
        dout[:, :M][keep] = src[keep]
        nz = L > 0                                                   # empty options get no <END> (:306-315)
        dout[np.arange(n)[nz], L[nz]] = self.word2ind['<END>']
        d['opt_len'] = L + 1
        d['opt_in'], d['opt_out'] = din, dout

    def processHistory(self, dtype):
        """dataloader.lua:203-278: round 1 = caption (first maxQ+maxA tokens); round r = Q(r-1)+A(r-1), or the
        running <END>-separated concatenation when concatHistory; then right-align."""
        d = self.data[dtype]
        cap, capL = d['cap'], d['cap_len']
        q, qL, a, aL = d['ques'], d['ques_len'], d['ans'], d['ans_len']
        n, R, MQ = q.shape
        MA = a.shape[2]
        W0 = MQ + MA
        if self.concatHistory:
            self.maxHistoryLen = min(R * W0, 300)
            W = self.maxHistoryLen
        else:
            W = W0
        hist = np.zeros((n, R, W), np.int64)
        hl = np.zeros((n, R), np.int64)
        END = self.word2ind['<END>']
        for i in range(n):
            lenH = 0
            for r in range(R):
                if r == 0:
                    hist[i, 0, :W0] = cap[i, :W0]
                    lenH = min(int(capL[i]), W0)
                else:
                    lq, la = int(qL[i, r - 1]), int(aL[i, r - 1])
                    if self.concatHistory:
                        hist[i, r, :lenH] = hist[i, r - 1, :lenH]
                        hist[i, r, lenH] = END
                        hist[i, r, lenH + 1:lenH + 1 + lq] = q[i, r - 1, :lq]
                        hist[i, r, lenH + 1 + lq:lenH + 1 + lq + la] = a[i, r - 1, :la]
                        lenH = lenH + lq + la + 1
                    else:
                        hist[i, r, :lq] = q[i, r - 1, :lq]
                        hist[i, r, lq:lq + la] = a[i, r - 1, :la]
                        lenH = lq + la
                hl[i, r] = lenH
        d['hist'] = rightAlign(hist, hl)
        d['hist_len'] = hl

    # -------------------------------------------------------------------------------- batches
    def getTrainBatch(self, params, batchSize=None):
        """dataloader.lua:324-339: `size` thread ids sampled WITH replacement"""
        size = int(batchSize or params['batchSize'])
        inds = self.rng.randint(1, int(params.get('numTrainThreads', self.numThreads['train'])) + 1, size=size)
        out = self.getIndexData(inds, params, 'train')
        if params['decoder'] == 'disc':
            o = self.getIndexOption(inds, params, 'train')
            out['options'] = o.reshape(o.shape[0] * o.shape[1], o.shape[2], -1)
            out['answer_ind'] = out['answer_ind'].reshape(-1)
        return out

    def getTestBatch(self, startId, params, dtype):
        """dataloader.lua:342-375"""
        nxt = min(self.numThreads[dtype] + 1, startId + int(params['batchSize']))
        inds = np.arange(startId, nxt)
        out = self.getIndexData(inds, params, dtype)
        o = self.getIndexOption(inds, params, dtype)
        if params['decoder'] == 'disc':
            out['options'] = o.reshape(o.shape[0] * o.shape[1], o.shape[2], -1)
            if dtype != 'test':
                out['answer_ind'] = out['answer_ind'].reshape(-1)
        else:
            out.update(o)
        out['num_rounds'] = self.data[dtype]['num_rounds'][inds - 1]
        return out, nxt

    def getIndexData(self, inds, params, dtype):
        """dataloader.lua:378-432 (inds are 1-based thread ids)"""
        d = self.data[dtype]
        ix = np.asarray(inds, np.int64) - 1
        out = {}
        mq = int(d['ques_len'][ix].max())
        out['ques_fwd'] = np.ascontiguousarray(d['ques_fwd'][ix][:, :, d['ques_fwd'].shape[2] - mq:]).astype(np.int32)
        if self.useHistory:
            mh = min(int(d['hist_len'][ix].max()), self.maxHistoryLen)
            out['hist'] = np.ascontiguousarray(d['hist'][ix][:, :, d['hist'].shape[2] - mh:]).astype(np.int32)
        if self.useIm:
            out['img_feat'] = d['img_fv'][d['img_pos'][ix] - 1]
        ma = int(d['ans_len'][ix].max())
        out['answer_in'] = np.ascontiguousarray(d['ans_in'][ix][:, :, :ma]).astype(np.int32)
        out['answer_out'] = np.ascontiguousarray(d['ans_out'][ix][:, :, :ma]).astype(np.int32)
        if dtype != 'test':
            out['answer_ind'] = d['ans_ind'][ix].astype(np.int32)
        return out

    def getIndexOption(self, inds, params, dtype):
        """dataloader.lua:435-477"""
        d = self.data[dtype]
        ix = np.asarray(inds, np.int64) - 1
        oi = d['opt'][ix]                                             # [B, R, 100] 1-based rows of opt_list
        flat = oi.reshape(-1) - 1
        if params['decoder'] == 'gen':
            ml = int(d['opt_len'][flat].max())
            shp = oi.shape + (-1,)
            return {'option_in': np.ascontiguousarray(d['opt_in'][flat].reshape(shp)[..., :ml]).astype(np.int32),
                    'option_out': np.ascontiguousarray(d['opt_out'][flat].reshape(shp)[..., :ml]).astype(np.int32)}
        return d['opt_list'][flat].reshape(oi.shape + (-1,)).astype(np.int32)
