#!/usr/bin/env python
"""evaluate.py -- counterpart of the reference's evaluate.lua (flags :16-30): loads a checkpoint written
by train.py, rebuilds the model from the SAVED modelParams (evaluate.lua:58-68), and ranks the 100
candidate answers of every round: -useGt 1 -> retrieve (R@1/5/10, median/mean rank, MRR), else predict.
Optionally dumps {image_id, round_id, ranks} records as JSON (evaluate.lua:104-107)."""
import argparse

import torch

from visdial_amd import opts, utils
from visdial_amd.dataloader import SyntheticDataloader
from visdial_amd.model import Model
from visdial_amd.checkpoint import load_checkpoint, restore_weights


def main():
    ap = argparse.ArgumentParser(description='Evaluate the Visual Dialog model')
    ap.add_argument('-loadPath', '--loadPath', required=True)
    ap.add_argument('-split', '--split', default='val')
    ap.add_argument('-useGt', '--useGt', type=int, default=1)
    ap.add_argument('-batchSize', '--batchSize', type=int, default=20)
    ap.add_argument('-gpuid', '--gpuid', type=int, default=0)
    ap.add_argument('-saveRanks', '--saveRanks', type=int, default=0)
    ap.add_argument('-saveRankPath', '--saveRankPath', default='logs/ranks.json')
    ap.add_argument('--numThreads', type=int, default=100)
    a = ap.parse_args()
    saved = load_checkpoint(a.loadPath)
    p = opts.derive(saved['modelParams'])
    p['gpuid'], p['batchSize'] = a.gpuid, a.batchSize
    dl = SyntheticDataloader(p, seed=4321, num_threads=a.numThreads)
    model = Model(p)
    restore_weights(model, saved)          # evaluate.lua:91
    if a.useGt:
        metrics, records = model.retrieve(dl, a.split)
    else:
        records = model.predict(dl, a.split)
    if a.saveRanks:
        utils.writeJSON(a.saveRankPath, [{k: (v.tolist() if hasattr(v, 'tolist') else v) for k, v in r.items()}
                                         for r in records])


if __name__ == '__main__':
    main()
