"""Generates the committed golden vectors under tests/golden/ from the CPU oracle.

The reference has no golden vectors of its own and cannot run in the build container (no Lua/Torch7),
so these are ORACLE outputs (fp64) for seeded inputs: they freeze the restated semantics (any later
edit of the oracle that changes a number fails tests/test_golden.py) and give the GPU path a fixed,
file-based target.  Data only: inputs (parameters, batch, dropout masks) and expected outputs
(loss, every gradient tensor, scores/ranks/metrics).        Run:  python tests/golden/make_golden.py
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


def main():
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
