#!/usr/bin/env python
"""generate.py -- counterpart of the reference's generate.lua: load a checkpoint written by train.py, run
beam search (default) or temperature sampling with the generative decoder over the first `maxThreads`
dialogs of the val split and write vis/results/results.json-style output ({opts, data}).
Needs the real data files (or their .npz twins): question text comes from the dataset vocabulary."""
import argparse
import os

import torch

from visdial_amd import opts, utils
from visdial_amd.dataloader import Dataloader
from visdial_amd.model import Model
from visdial_amd.checkpoint import load_checkpoint, restore_weights


def main():
    ap = argparse.ArgumentParser(description='Test the VisDial model for generation')
    ap.add_argument('-inputImg', '--inputImg', default='data/data_img.h5')
    ap.add_argument('-inputQues', '--inputQues', default='data/visdial_data.h5')
    ap.add_argument('-inputJson', '--inputJson', default='data/visdial_params.json')
    ap.add_argument('-loadPath', '--loadPath', required=True)
    ap.add_argument('-paramOrder', '--paramOrder', default='', help="layout of the .t7 flat vector: '' | declaration | <json> (visdial_amd/t7.py resolve_order)")
    ap.add_argument('-resultPath', '--resultPath', default='vis/results')
    ap.add_argument('-beamSize', '--beamSize', type=int, default=5)
    ap.add_argument('-beamLen', '--beamLen', type=int, default=20)
    ap.add_argument('-sampleWords', '--sampleWords', type=int, default=0)
    ap.add_argument('-temperature', '--temperature', type=float, default=1.0)
    ap.add_argument('-maxThreads', '--maxThreads', type=int, default=50)
    ap.add_argument('-gpuid', '--gpuid', type=int, default=0)
    ap.add_argument('-host', '--host', default='python', choices=['python', 'native'],
                    help="'native' drives the model-level C ABI (what lua/model.lua calls)")
    a = vars(ap.parse_args())
    saved = load_checkpoint(a['loadPath'])
    p = opts.derive(saved['modelParams'])                      # generate.lua:57-70
    p['gpuid'] = a['gpuid']
    p.update(inputImg=a['inputImg'], inputQues=a['inputQues'], inputJson=a['inputJson'])
    # generate.lua:57-70 derives useHistory / useIm for the dataloader but NOT concatHistory (train.lua and evaluate.lua do): the history
    # of a generation run is the previous round's question + answer even for the lf-* encoders, with the default maxHistoryLen.
    # Reproduced as is (found by executing generate.lua: tests/golden/make_reference_train_golden.py).
    dl = Dataloader(seed=1234).initialize(dict(p, concatHistory=False, maxHistoryLen=60), ['val'])
    for k in ('vocabSize', 'maxQuesCount', 'maxQuesLen', 'maxAnsLen'):
        p[k] = getattr(dl, k)
    if a['host'] == 'native':
        from visdial_amd.native import NativeModel
        model = NativeModel(p)
    else:
        model = Model(p)
    restore_weights(model, saved, a['paramOrder'] or None)
    answers = model.generateAnswers(dl, 'val', dict(beamSize=a['beamSize'], beamLen=a['beamLen'],
                                                    maxThreads=a['maxThreads'], sampleWords=a['sampleWords'],
                                                    temperature=a['temperature']))
    os.makedirs(a['resultPath'], exist_ok=True)
    path = os.path.join(a['resultPath'], 'results.json')
    utils.writeJSON(path, {'opts': a, 'data': answers})
    print('Writing the results to ' + path)


if __name__ == '__main__':
    main()
