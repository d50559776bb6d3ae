"""Generates the committed golden vectors under tests/golden/ from the CPU oracle.

The reference has no golden vectors of its own and cannot run in the build container (no Lua/Torch7),
so these are ORACLE outputs (fp64) for seeded inputs: they freeze the restated semantics (any later
edit of the oracle that changes a number fails tests/test_golden.py) and give the GPU path a fixed,
file-based target.  Data only: inputs (parameters, batch, dropout masks) and expected outputs
(loss, every gradient tensor, scores/ranks/metrics).        Run:  python tests/golden/make_golden.py

`python tests/golden/make_golden.py --full` writes the FULL-SIZE fixture of BASELINE.json configs[3] (SURVEY.md 8c:
"one at full B=20 stored as checksums/slices"): see full_case() / full_outputs() below.  ~20 GB of RAM, minutes.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

from conftest import small_params                      # noqa: E402
from oracle import visdial_oracle as vo                # noqa: E402
from visdial_amd.dataloader import SyntheticDataloader  # noqa: E402
from visdial_amd.opts import derive                    # noqa: E402

CASES = [   # the four configured encoder/decoder pairs of BASELINE.json:configs (reduced sizes)
    ('lf-ques', 'gen', dict(dropout=0.5, imgNorm=1, batchSize=2)),
    ('lf-ques-im-hist', 'gen', dict(dropout=0.5, imgNorm=1, batchSize=2)),
    ('hre-ques-im-hist', 'disc', dict(imgNorm=1, batchSize=2)),
    ('mn-att-ques-im-hist', 'disc', dict(batchSize=2)),
]


def masks_for(p, batch, rng):
    enc = p['encoder']
    B, R, Tq = batch['ques_fwd'].shape
    N, H, E = B * R, p['rnnHiddenSize'], p['embedSize']
    if enc == 'lf-ques':
        shp = dict(fuse=(N, H))
    elif enc == 'lf-ques-im-hist':
        shp = dict(fuse=(N, 2 * H + p['imgFeatureSize']))
    elif enc == 'mn-att-ques-im-hist':
        Th = batch['hist'].shape[2]
        S2, K = p['imgSpatialSize'] ** 2, p['commonEmbeddingSize']
        shp = dict(q_emb=(Tq, N, E), h_emb=(Th, N, E), hatt=(N, H), img_tr=(N, S2, H), iqc=(N, S2, K), u=(N, H))
    else:
        return None
    pd = p.get('dropout', 0.5) if enc.startswith('lf') else 0.5
    return {k: (rng.rand(*s) > pd).astype(np.uint8) for k, s in shp.items()}


# ---------------------------------------------------------------------------------------------------------------
# Full-size fixture: mn-att-ques-im-hist + disc, 20 dialogs x 10 rounds x 100 options, 14x14x512, V = 11 322, H = 512,
# dropout ON with pinned masks (model.lua:249-342, decoders/disc.lua:3-32, utils.lua:106-160).  Inputs are
# seed-reproducible (numpy RandomState / default_rng), so the file stores only their SHA-256 -- a drift of a generator
# is reported as such, not as a parity failure -- and the oracle's fp64 OUTPUTS: loss, the whole [200 x 100] score
# matrix (training-mode forward and evaluate-mode forward), ranks, R@k / MRR, and for each of the 14.2 M gradient
# values' tensors: L2 norm, sum, a strided sample of <= 16 384 entries, and a 64-row random-sign sketch
# <g, r_j> (E[<d, r>^2] = |d|^2, so the sketch of a difference estimates the FULL tensor's L2 error).
FULL_NAME = 'full__mn-att-ques-im-hist__disc.npz'
SAMPLE, SKETCH = 16384, 64
# the full-size fixtures: BASELINE.json configs[3] (the headline), configs[2] (hre-ques-im-hist + disc on fc7 features; its encoder has
# no pinned-mask sites in masks_for, so the fixture is the evaluate-mode step: no dropout) and the SHAPE of configs[4] (7x7x2048 ResNet-200
# map; the fp64 outputs are what its bf16 pass is held to within its stated bound, and the fp32-grade passes within 1e-4)
FULL_CASES = {
    'mn-att': (FULL_NAME, dict(encoder='mn-att-ques-im-hist', decoder='disc', imgFeatureSize=512, imgSpatialSize=14)),
    'hre': ('full__hre-ques-im-hist__disc.npz', dict(encoder='hre-ques-im-hist', decoder='disc', imgFeatureSize=4096)),
    'mn-att-7x7': ('full__mn-att-7x7x2048__disc.npz', dict(encoder='mn-att-ques-im-hist', decoder='disc', imgFeatureSize=2048, imgSpatialSize=7)),
    # configs[1]: the generative decoder over the concatenated history (250 steps at this seed); no option scores: loss + gradients only
    'lf-gen': ('full__lf-ques-im-hist__gen.npz', dict(encoder='lf-ques-im-hist', decoder='gen', imgFeatureSize=4096)),
}


def full_case(name='mn-att'):
    """(p, batch, masks, P) of a full-size fixture; every array deterministic in the seeds below"""
    from visdial_amd.opts import default_params
    p = default_params(batchSize=20, vocabSize=11322, gpuid=0, maxHistoryLenPerRound=40, **FULL_CASES[name][1])
    batch = SyntheticDataloader(p, seed=1234, fast=True).getTrainBatch(p)
    masks = masks_for(p, batch, np.random.RandomState(5))
    P = vo.init_params(p['encoder'], p['decoder'], p, seed=77, dtype=np.float32)
    return p, batch, masks, P


def digest(d):
    import hashlib
    h = hashlib.sha256()
    for k in sorted(d):
        a = np.ascontiguousarray(d[k])
        h.update(k.encode()); h.update(str(a.dtype).encode()); h.update(str(a.shape).encode()); h.update(a.tobytes())
    return h.hexdigest()


def sample_index(n):
    """<= SAMPLE entries of a flat tensor: a fixed stride over the whole tensor"""
    return np.arange(0, n, max(1, n // SAMPLE))[:SAMPLE]


def sketch(name, flat):
    """64 random-sign projections of a flat fp64 vector (signs seeded by the tensor name and size)"""
    import zlib
    rng = np.random.RandomState(zlib.crc32(name.encode()) & 0x7fffffff)
    out = np.empty(SKETCH)
    for j in range(SKETCH):
        r = rng.randint(0, 2, flat.size).astype(np.int8) * 2 - 1
        out[j] = float(np.dot(flat, r))
    return out


def full_outputs(name='mn-att'):
    p, batch, masks, P = full_case(name)
    P64 = {k: v.astype(np.float64) for k, v in P.items()}
    drop = {k: v.astype(np.float64) for k, v in masks.items()} if masks else None
    out = {'digest.params': np.array(digest(P)), 'digest.batch': np.array(digest(batch)),
           'digest.masks': np.array(digest(masks or {}))}
    r = vo.forward_backward(p['encoder'], p['decoder'], P64, p, batch, drop)
    disc = r['scores'] is not None
    gt = batch['answer_ind'].reshape(-1) - 1 if disc else None
    out['loss'] = np.float64(r['loss'])
    if disc:
        out['scores'] = r['scores']
        out['gt_ranks'] = vo.compute_ranks(r['scores'], gt)
    for k, g in r['grads'].items():
        f = g.reshape(-1)
        out['gnorm.' + k] = np.float64(np.linalg.norm(f))
        out['gsum.' + k] = np.float64(f.sum())
        out['gsample.' + k] = f[sample_index(f.size)].copy()
        out['gsketch.' + k] = sketch(k, f)
    del r
    ev = vo.forward_backward(p['encoder'], p['decoder'], P64, p, batch, None, only_forward=True)   # evaluate(): no dropout
    out['eval.loss'] = np.float64(ev['loss'])
    if not disc:
        return out
    out['eval.scores'] = ev['scores']
    out['eval.ranks'] = vo.compute_ranks(ev['scores']).astype(np.int16)
    out['eval.gt_ranks'] = vo.compute_ranks(ev['scores'], gt)
    m = vo.process_ranks(out['eval.gt_ranks'])
    out['eval.metrics'] = np.array([m[k] for k in ('r@1', 'r@5', 'r@10', 'medianR', 'meanR', 'meanRR')])
    return out


def main_full():
    import time
    names = [a for a in sys.argv[1:] if a in FULL_CASES] or list(FULL_CASES)       # `--full hre mn-att-7x7`: only these
    for name in names:
        t0 = time.time()
        out = full_outputs(name)
        path = os.path.join(HERE, FULL_CASES[name][0])
        np.savez_compressed(path, **out)
        print('%s  loss %.12f  eval loss %.12f  %d arrays  %.1f KB  (%.0f s)' % (
            FULL_CASES[name][0], out['loss'], out['eval.loss'], len(out), os.path.getsize(path) / 1024.0, time.time() - t0), flush=True)


def main():
    if '--full' in sys.argv:
        return main_full()
    for enc, dec, kw in CASES:
        p = derive(small_params(encoder=enc, decoder=dec, **kw))
        dl = SyntheticDataloader(p, seed=2024)
        batch = dl.getTrainBatch(p)
        P = vo.init_params(enc, dec, p, seed=77, dtype=np.float32)          # fp32-representable parameters
        P64 = {k: v.astype(np.float64) for k, v in P.items()}
        masks = masks_for(p, batch, np.random.RandomState(5))
        drop = {k: v.astype(np.float64) for k, v in masks.items()} if masks else None
        r = vo.forward_backward(enc, dec, P64, p, batch, drop)
        out = {'loss': np.float64(r['loss'])}
        for k, v in P.items():
            out['param.' + k] = v
        for k, v in batch.items():
            out['batch.' + k] = v
        for k, v in (masks or {}).items():
            out['mask.' + k] = v
        for k, v in r['grads'].items():
            out['grad.' + k] = v.astype(np.float64)
        if dec == 'disc':
            ev = vo.forward_backward(enc, dec, P64, p, batch, None, only_forward=True)      # evaluate(): no dropout
            out['eval.scores'] = ev['scores']
            out['eval.ranks'] = vo.compute_ranks(ev['scores'])
            out['eval.gt_ranks'] = vo.compute_ranks(ev['scores'], batch['answer_ind'] - 1)
            m = vo.process_ranks(out['eval.gt_ranks'])
            out['eval.metrics'] = np.array([m[k] for k in ('r@1', 'r@5', 'r@10', 'medianR', 'meanR', 'meanRR')])
        path = os.path.join(HERE, '%s__%s.npz' % (enc, dec))
        np.savez_compressed(path, **out)
        print('%-40s loss %.12f  %d arrays  %.1f KB' % (os.path.basename(path), r['loss'], len(out),
                                                       os.path.getsize(path) / 1024.0))


if __name__ == '__main__':
    main()
