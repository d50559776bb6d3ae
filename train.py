#!/usr/bin/env python
"""train.py -- counterpart of the reference's train.lua (same flags: visdial_amd/opts.py <-> opts.lua).

  python train.py -encoder mn-att-ques-im-hist -decoder disc -imgFeatureSize 512 -batchSize 20 --maxIters 300

Data is synthetic with the reference dataloader's exact layout (no VisDial HDF5/JSON offline).
Loop, logging line, lr schedule, checkpoint cadence and resume semantics follow train.lua:76-121:
resume restores weights + learning rate only (not the Adam moments, not the iteration counter)."""
import math
import os
import time

import numpy as np
import torch

from visdial_amd import opts
from visdial_amd.dataloader import Dataloader, SyntheticDataloader
from visdial_amd.model import Model
from visdial_amd.checkpoint import load_checkpoint, restore_weights, save_t7


def _plain(d):
    """options as python primitives only (checkpoints are read back with torch.load(weights_only=True))"""
    out = {}
    for k, v in d.items():
        if isinstance(v, (bool, str)) or v is None:
            out[k] = v
        elif isinstance(v, (int, np.integer)):
            out[k] = int(v)
        elif isinstance(v, (float, np.floating)):
            out[k] = float(v)
    return out


def main():
    opt = opts.parse()
    print(opt)
    np.random.seed(1234)                                         # train.lua:12
    saved = None
    mp = opt                                                     # `modelParams = opt` (train.lua:26)
    if opt['loadPath']:
        saved = load_checkpoint(opt['loadPath'])                 # train.lua:32-41
        mp = saved['modelParams']
        for k in ('imgNorm', 'encoder', 'decoder'):              # the only three options taken from the checkpoint
            opt[k] = mp[k]
        mp['gpuid'], mp['batchSize'] = opt['gpuid'], opt['batchSize']
        # run control and runtime (non-architectural) choices stay with the command line
        for k in ('numEpochs', 'maxIters', 'savePath', 'saveIter', 'host', 'lstmPrecision', 'saveFormat',
                  'synthetic'):
            mp[k] = opt.get(k)
    # the dataloader is built from the CURRENT command line (train.lua:47-48), never from paths stored in a checkpoint
    have = lambda p: os.path.exists(p) or os.path.exists(p[:-3] + '.npz')
    if os.path.exists(opt['inputJson']) and have(opt['inputQues']):
        dataloader = Dataloader(seed=1234).initialize(opt, ['train'])            # real VisDial files
    elif opt['loadPath'] and not opt.get('synthetic'):
        raise SystemExit('-loadPath given but no dataset at %s / %s: refusing to resume on synthetic data '
                         '(pass -synthetic 1 to force)' % (opt['inputJson'], opt['inputQues']))
    else:
        print('no dataset at %s: using synthetic VisDial-shaped batches' % opt['inputQues'])
        dataloader = SyntheticDataloader(opt, seed=1234, num_threads=opt['numTrainThreads'])
    opt = mp
    for k in ('vocabSize', 'maxQuesCount', 'maxQuesLen', 'maxAnsLen'):   # train.lua:55-59
        opt[k] = getattr(dataloader, k)
    opt['numTrainThreads'] = dataloader.numTrainThreads
    opt['numOptions'] = getattr(dataloader, 'numOptions', 100)
    os.makedirs(opt['savePath'], exist_ok=True)
    opt['numIterPerEpoch'] = int(math.ceil(opt['numTrainThreads'] / float(opt['batchSize'])))
    print('\n%d iter per epoch.' % opt['numIterPerEpoch'])
    if opt.get('host', 'python') == 'native':     # model-level C ABI (the calls lua/model.lua makes)
        from visdial_amd.native import NativeModel
        model = NativeModel(opt)
    else:
        model = Model(opt)
    if saved is not None:                                        # train.lua:78-81
        restore_weights(model, saved, opt.get('paramOrder') or None)
        model.optims['learningRate'] = saved['optims']['learningRate']
    print('Training..')
    total = opt['numEpochs'] * opt['numIterPerEpoch']
    if opt.get('maxIters'):
        total = min(total, opt['maxIters'])
    t0 = time.time()
    for it in range(1, total + 1):
        model.trainIteration(dataloader)
        if it % (opt['saveIter'] * opt['numIterPerEpoch']) == 0:      # train.lua:95-102
            ep = it // opt['numIterPerEpoch']
            if opt.get('saveFormat', 't7') == 't7':
                save_t7(os.path.join(opt['savePath'], 'model_epoch_%d.t7' % ep), model, _plain(opt))
            else:
                torch.save({'modelW': model.wrapperW.cpu(), 'optims': {k: model.optims[k] for k in model.optims.keys()},
                            'modelParams': _plain(opt)},
                           os.path.join(opt['savePath'], 'model_epoch_%d.pt' % ep))
        if it % 100 == 0:                                            # train.lua:108-115
            torch.cuda.synchronize()
            getattr(model, 'synchronize', lambda: None)()
            rounds = 100 * opt['batchSize'] * opt['maxQuesCount']
            print('[%s][Epoch:%.02f][Iter:%d][Loss:%.05f][lr:%f][%.0f QA-rounds/s]' % (
                time.ctime(), it / float(opt['numIterPerEpoch']), it, model.runningLoss,
                model.optims['learningRate'], rounds / (time.time() - t0)))
            t0 = time.time()
    if opt.get('saveFormat', 't7') == 't7':                          # train.lua:120-121
        save_t7(os.path.join(opt['savePath'], 'model_final.t7'), model, _plain(opt))
    else:
        torch.save({'modelW': model.wrapperW.float().cpu(), 'modelParams': _plain(opt),
                    'optims': {k: model.optims[k] for k in model.optims.keys()}},
                   os.path.join(opt['savePath'], 'model_final.pt'))


if __name__ == '__main__':
    main()
