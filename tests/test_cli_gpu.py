"""train.py / evaluate.py round trip on the GPU box: a short training run lowers the loss, writes a
checkpoint, evaluate.py reloads it and prints the retrieval metrics."""
import os
import subprocess
import sys

import pytest
import torch

from conftest import ROOT

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("host", ['python', 'native'])
def test_train_then_evaluate(tmp_path, host):
    """host = which layer of the C ABI drives the library: operator-level (visdial_amd/model.py) or model-level
    (visdial_amd/native.py = the calls lua/model.lua makes); checkpoints are interchangeable (evaluated by the OTHER host)"""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    save = str(tmp_path / "ckpt") + "/"
    common = ['-encoder', 'mn-att-ques-im-hist', '-decoder', 'disc', '-imgFeatureSize', '64', '-imgSpatialSize', '4',
              '-rnnHiddenSize', '64', '-embedSize', '32', '-commonEmbeddingSize', '64', '-batchSize', '4',
              '--vocabSize', '100', '--numTrainThreads', '40', '-host', host]
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'train.py')] + common +
                       ['-savePath', save, '-numEpochs', '20', '-saveIter', '10', '--maxIters', '200'],
                       capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('[') and '[Loss:' in l]
    assert len(lines) == 2 and '[lr:' in lines[0]
    loss = [float(l.split('[Loss:')[1].split(']')[0]) for l in lines]
    assert loss[1] < loss[0] < 6.0
    # checkpoints under the reference's names, in the Torch7 binary format (train.lua:99-102,120-121)
    assert os.path.exists(save + 'model_final.t7') and os.path.exists(save + 'model_epoch_10.t7')
    from visdial_amd import t7
    ck = t7.load(save + 'model_epoch_10.t7')
    assert set(ck) >= {'modelW', 'optims', 'modelParams'} and ck['optims']['learningRate'] < 1e-3
    assert ck.get('vdLayout') == 'reference'                  # every encoder is written in the reference's getParameters() order, and says so
    e = subprocess.run([sys.executable, os.path.join(ROOT, 'evaluate.py'), '-loadPath', save + 'model_final.t7',
                        '-batchSize', '4', '--numThreads', '8', '-saveRanks', '1', '-saveRankPath',
                        str(tmp_path / 'ranks.json'), '-perplexity', '1', '-host', 'native' if host == 'python' else 'python'],
                       capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert e.returncode == 0, e.stdout[-2000:] + e.stderr[-2000:]
    assert 'r@1:' in e.stdout and 'meanRR:' in e.stdout and os.path.exists(str(tmp_path / 'ranks.json'))
    # resume from the epoch checkpoint (train.lua:32-41,78-81): weights + learning rate come back, the run continues
    # on the host named on the COMMAND LINE (not the one stored in the checkpoint)
    other = 'native' if host == 'python' else 'python'
    r2 = subprocess.run([sys.executable, os.path.join(ROOT, 'train.py')] + common[:-1] + [other] +
                        ['-loadPath', save + 'model_epoch_10.t7', '-synthetic', '1', '-savePath', save + 'resumed/',
                         '-numEpochs', '10', '-saveIter', '100', '--maxIters', '100', '-saveFormat', 'pt'],
                        capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r2.returncode == 0, r2.stdout[-2000:] + r2.stderr[-2000:]
    l2 = [float(l.split('[Loss:')[1].split(']')[0]) for l in r2.stdout.splitlines() if '[Loss:' in l]
    lr2 = [float(l.split('[lr:')[1].split(']')[0]) for l in r2.stdout.splitlines() if '[lr:' in l]
    assert len(l2) == 1 and l2[0] < loss[0]                 # it starts from trained weights, not from scratch
    assert lr2[0] < ck['optims']['learningRate']           # the decayed learning rate was restored and kept decaying
    assert os.path.exists(save + 'resumed/model_final.pt')


def test_train_evaluate_on_real_format_files(tmp_path):
    """train.py and evaluate.py on files in the reference's dataset format (prepro.py's dataset names, here as the
    .npz twins of visdial_data.h5 / data_img.h5 + visdial_params.json): the real `Dataloader` path end to end --
    training lowers the loss, evaluate.py (evaluate.lua:80-81: dataloader on the chosen split) prints retrieval
    metrics and perplexity-style loss and writes {image_id, round_id, ranks} records with the dataset's image ids and
    only the rounds that exist (model.lua:174-184)."""
    import json
    import numpy as np
    from test_dataloader_cpu import raw_dataset
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    rng = np.random.RandomState(11)
    n, R = 12, 4
    info, raw, img = raw_dataset(rng, n=n, R=R, MQ=6, MA=5, V=30, O=5, nopt=40, F=8, att=True)
    # a val split = the same dialogs under the val names, with ragged round counts
    for k in list(raw):
        raw[k.replace('_train', '_val')] = raw[k]
    raw['num_rounds_val'] = np.array([R, R - 1] * (n // 2), np.uint32)
    img['images_val'] = img['images_train']
    info['unique_img_val'] = ['VisualDialog_val2018_%012d.jpg' % (1000 + i) for i in range(n)]
    np.savez(str(tmp_path / 'visdial_data.npz'), **raw)
    np.savez(str(tmp_path / 'data_img.npz'), **img)
    json.dump(info, open(str(tmp_path / 'visdial_params.json'), 'w'))
    data = ['-inputQues', str(tmp_path / 'visdial_data.h5'), '-inputImg', str(tmp_path / 'data_img.h5'),
            '-inputJson', str(tmp_path / 'visdial_params.json')]
    save = str(tmp_path / "ckpt") + "/"
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'train.py'), '-encoder', 'mn-att-ques-im-hist', '-decoder', 'disc',
                        '-imgFeatureSize', '8', '-imgSpatialSize', '3', '-rnnHiddenSize', '32', '-embedSize', '16',
                        '-commonEmbeddingSize', '32', '-batchSize', '4', '-savePath', save, '-numEpochs', '100',
                        '-saveIter', '1000', '--maxIters', '200', '-saveFormat', 'pt'] + data,
                       capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert 'using synthetic' not in r.stdout
    loss = [float(l.split('[Loss:')[1].split(']')[0]) for l in r.stdout.splitlines() if '[Loss:' in l]
    assert len(loss) == 2 and loss[1] < loss[0]
    ranks = str(tmp_path / 'logs' / 'ranks.json')
    e = subprocess.run([sys.executable, os.path.join(ROOT, 'evaluate.py'), '-loadPath', save + 'model_final.pt',
                        '-batchSize', '5', '-split', 'val', '-saveRanks', '1', '-saveRankPath', ranks,
                        '-perplexity', '1'] + data, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert e.returncode == 0, e.stdout[-2000:] + e.stderr[-2000:]
    assert 'SYNTHETIC' not in e.stdout and 'r@1:' in e.stdout and 'Perplexity' in e.stdout
    rec = json.load(open(ranks))
    assert len(rec) == int(raw['num_rounds_val'].sum())                       # only the rounds that exist
    assert rec[0]['image_id'] == 1000 and rec[0]['round_id'] == 1             # tonumber(match(v, '000%d+'))
    assert all(1 <= x['ranks'] <= 5 for x in rec)
    # predict on the same split: all candidate ranks per round
    e = subprocess.run([sys.executable, os.path.join(ROOT, 'evaluate.py'), '-loadPath', save + 'model_final.pt',
                        '-batchSize', '5', '-split', 'val', '-useGt', '0', '-saveRanks', '1', '-saveRankPath', ranks] + data,
                       capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert e.returncode == 0, e.stdout[-2000:] + e.stderr[-2000:]
    rec = json.load(open(ranks))
    assert sorted(rec[0]['ranks']) == [1.0, 2.0, 3.0, 4.0, 5.0]
