#!/usr/bin/env python
"""evaluate.py -- counterpart of the reference's evaluate.lua (flags :16-30): loads a checkpoint written by train.py
(or a reference .t7), rebuilds the model from the SAVED modelParams (evaluate.lua:58-68), initialises the dataloader
on the chosen split of the REAL data files (evaluate.lua:80-81) and ranks the 100 candidate answers of every round:
-useGt 1 -> retrieve (R@1/5/10, median/mean rank, MRR), else predict.  Optionally dumps the {image_id, round_id,
ranks} records as JSON (evaluate.lua:104-107; the EvalAI submission format).  Without data files on disk it falls
back to synthetic VisDial-shaped batches (plumbing check only; says so)."""
import argparse
import os

from visdial_amd import opts, utils
from visdial_amd.checkpoint import load_checkpoint, restore_weights
from visdial_amd.dataloader import Dataloader, SyntheticDataloader
from visdial_amd.model import Model


def main():
    ap = argparse.ArgumentParser(description='Evaluate the Visual Dialog model')
    ap.add_argument('-inputImg', '--inputImg', default='data/data_img.h5')
    ap.add_argument('-inputQues', '--inputQues', default='data/visdial_data.h5')
    ap.add_argument('-inputJson', '--inputJson', default='data/visdial_params.json')
    ap.add_argument('-loadPath', '--loadPath', required=True)
    ap.add_argument('-paramOrder', '--paramOrder', default='', help="layout of the .t7 flat vector: '' | declaration | <json> (visdial_amd/t7.py resolve_order)")
    ap.add_argument('-split', '--split', default='val')
    ap.add_argument('-useGt', '--useGt', type=int, default=1)
    ap.add_argument('-batchSize', '--batchSize', type=int, default=20)
    ap.add_argument('-gpuid', '--gpuid', type=int, default=0)
    ap.add_argument('-saveRanks', '--saveRanks', type=int, default=0)
    ap.add_argument('-saveRankPath', '--saveRankPath', default='logs/ranks.json')
    ap.add_argument('-perplexity', '--perplexity', type=int, default=0, help='also run Model:evaluate (model.lua:109-139)')
    ap.add_argument('--numThreads', type=int, default=100, help='synthetic fallback only')
    ap.add_argument('-host', '--host', default='python', choices=['python', 'native'],
                    help="'native' drives the model-level C ABI (what lua/model.lua calls)")
    a = ap.parse_args()
    saved = load_checkpoint(a.loadPath)
    p = opts.derive(saved['modelParams'])                    # sets useHistory / useIm / concatHistory (evaluate.lua:69-75)
    p['gpuid'], p['batchSize'], p['useGt'] = a.gpuid, a.batchSize, bool(a.useGt)
    p.update(inputImg=a.inputImg, inputQues=a.inputQues, inputJson=a.inputJson)
    have = lambda f: os.path.exists(f) or os.path.exists(f[:-3] + '.npz')
    if os.path.exists(a.inputJson) and have(a.inputQues):
        dl = Dataloader(seed=1234).initialize(p, [a.split])                      # evaluate.lua:80-81
        for k in ('vocabSize', 'maxQuesCount', 'maxQuesLen', 'maxAnsLen', 'numOptions'):
            p[k] = getattr(dl, k)
    else:
        print('no dataset at %s: ranking SYNTHETIC batches (plumbing check, the metrics mean nothing)' % a.inputQues)
        dl = SyntheticDataloader(p, seed=4321, num_threads=a.numThreads)
    if a.host == 'native':
        from visdial_amd.native import NativeModel
        model = NativeModel(p)
    else:
        model = Model(p)
    restore_weights(model, saved, a.paramOrder or None)          # evaluate.lua:91
    print('Evaluating..')
    if a.perplexity:
        model.evaluate(dl, a.split)
    if a.useGt:
        metrics, records = model.retrieve(dl, a.split)
    else:
        records = model.predict(dl, a.split)
    if a.saveRanks:
        print('Writing ranks to %s' % a.saveRankPath)
        os.makedirs(os.path.dirname(os.path.abspath(a.saveRankPath)), exist_ok=True)
        utils.writeJSON(a.saveRankPath, records)


if __name__ == '__main__':
    main()
