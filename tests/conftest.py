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
