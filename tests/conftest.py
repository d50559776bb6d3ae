import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _cpu_quota():
    try:
        q, per = open('/sys/fs/cgroup/cpu.max').read().split()
        return None if q == 'max' else float(q) / float(per)
    except Exception:
        return None


# The GPU boxes show 256 logical CPUs but run the container under a 16-CPU quota: BLAS / OpenMP pools sized by
# cpu_count are throttled to a fraction of what 2 x quota threads reach (the numpy fp64 oracle dominates the -m gpu
# suite's wall time).  Size the pools to the quota; an explicit OMP_NUM_THREADS wins.
_q = _cpu_quota()
if _q and 'OMP_NUM_THREADS' not in os.environ:
    _n = str(max(1, int(round(2 * _q))))
    for _k in ('OMP_NUM_THREADS', 'OPENBLAS_NUM_THREADS', 'MKL_NUM_THREADS'):
        os.environ.setdefault(_k, _n)                 # for libraries not loaded yet
    try:
        from threadpoolctl import threadpool_limits
        threadpool_limits(limits=int(_n))             # for the ones that already are
    except Exception:
        pass


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def small_params(**kw):
    """Reduced-size hyper-parameters for oracle-speed parity cases (same keys as opts.lua)."""
    p = dict(encoder='mn-att-ques-im-hist', decoder='disc', vocabSize=50, embedSize=12, rnnHiddenSize=32,
             numLayers=2, dropout=0.5, imgFeatureSize=16, imgEmbedSize=12, imgSpatialSize=3,
             commonEmbeddingSize=32, numAttentionLayers=1, maxQuesCount=4, maxQuesLen=6, maxAnsLen=5,
             maxHistoryLenPerRound=8, numOptions=7, batchSize=2, imgNorm=0, learningRate=1e-3,
             lrDecayRate=0.9997592083, minLRate=5e-5)
    p.update(kw)
    return p


@pytest.fixture
def params_small():
    return small_params()


def grad_mismatches(g, ref, tol=1e-4, zero=()):
    """[(rel-L2 error, tensor name)] of every gradient tensor that misses the bound.  The bound is RELATIVE (north_star:
    1e-4 in fp32) for every tensor with a gradient; only a tensor whose reference gradient is identically zero
    (|ref| < 1e-12: e.g. 'att.b', which feeds a softmax over the regions, or the weights of a branch that saw only
    padding) gets an absolute bound instead -- a relative error is undefined there."""
    import numpy as np
    bad = []
    for k in ref:
        a, b = np.asarray(g[k], np.float64), np.asarray(ref[k], np.float64)
        nb = float(np.linalg.norm(b))
        if nb < 1e-12 or k in zero:        # `zero`: tensors whose EXACT gradient is 0 when the reference is itself fp32
            if np.abs(a).max() >= 1e-6 or np.abs(b).max() >= 1e-6:
                bad.append((float(np.abs(a).max()), k + ' (reference gradient is zero)'))
            continue
        err = float(np.linalg.norm(a - b)) / nb
        if err >= tol:
            bad.append((err, k))
    return bad


def unexplained_rank_flips(dev_scores, ref_scores, tol=1e-4):
    """(flipped pairs, unexplained ones): option pairs whose ORDER differs between the device scores and the fp64 oracle's
    scores; a flip is explained only if the fp64 scores are a near tie, |s_a - s_b| < tol * max(1, |s_a|, |s_b|).  Ties
    break by lower index first on both sides (utils.lua:106-128)."""
    import numpy as np
    dev, ref = np.asarray(dev_scores, np.float64), np.asarray(ref_scores, np.float64)
    idx = np.arange(ref.shape[1])
    flipped, bad = 0, []
    before = lambda s: (s[:, None] > s[None, :]) | ((s[:, None] == s[None, :]) & (idx[:, None] < idx[None, :]))
    for r in range(ref.shape[0]):
        f = before(dev[r]) != before(ref[r])
        if not f.any():
            continue
        a, b = np.nonzero(np.triu(f, 1))
        flipped += a.size
        margin = np.abs(ref[r, a] - ref[r, b])
        scale = np.maximum(1.0, np.maximum(np.abs(ref[r, a]), np.abs(ref[r, b])))
        bad += [(r, int(a[k]), int(b[k]), float(margin[k])) for k in np.nonzero(margin >= tol * scale)[0]]
    return flipped, bad
