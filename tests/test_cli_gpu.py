"""train.py / evaluate.py round trip on the GPU box: a short training run lowers the loss, writes a
checkpoint, evaluate.py reloads it and prints the retrieval metrics."""
import os
import subprocess
import sys

import pytest
import torch

from conftest import ROOT

pytestmark = pytest.mark.gpu


def test_train_then_evaluate(tmp_path):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    save = str(tmp_path / "ckpt") + "/"
    common = ['-encoder', 'mn-att-ques-im-hist', '-decoder', 'disc', '-imgFeatureSize', '64', '-imgSpatialSize', '4',
              '-rnnHiddenSize', '64', '-embedSize', '32', '-commonEmbeddingSize', '64', '-batchSize', '4',
              '--vocabSize', '100', '--numTrainThreads', '40']
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'train.py')] + common +
                       ['-savePath', save, '-numEpochs', '20', '-saveIter', '10', '--maxIters', '200'],
                       capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('[') and '[Loss:' in l]
    assert len(lines) == 2 and '[lr:' in lines[0]
    loss = [float(l.split('[Loss:')[1].split(']')[0]) for l in lines]
    assert loss[1] < loss[0] < 6.0
    assert os.path.exists(save + 'model_final.pt') and os.path.exists(save + 'model_epoch_10.pt')
    e = subprocess.run([sys.executable, os.path.join(ROOT, 'evaluate.py'), '-loadPath', save + 'model_final.pt',
                        '-batchSize', '4', '--numThreads', '8', '-saveRanks', '1', '-saveRankPath',
                        str(tmp_path / 'ranks.json')], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert e.returncode == 0, e.stdout[-2000:] + e.stderr[-2000:]
    assert 'r@1:' in e.stdout and 'meanRR:' in e.stdout and os.path.exists(str(tmp_path / 'ranks.json'))
