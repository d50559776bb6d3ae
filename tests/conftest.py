import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


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
