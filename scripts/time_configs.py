"""Step time of every BASELINE.json configuration that fits one GPU (configs[1..4]), through both hosts, with a fresh
synthetic batch every step (pipelined trainIteration).  configs[3] is the bench line; the others are parity-test cases
and these timings are informative only."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from visdial_amd.dataloader import SyntheticDataloader
from visdial_amd.model import Model
from visdial_amd.native import NativeModel
from visdial_amd.opts import default_params

CONFIGS = [
    ("configs[1] lf-ques-im-hist + gen, fc7 4096, batch 20", dict(encoder='lf-ques-im-hist', decoder='gen', imgFeatureSize=4096, batchSize=20)),
    ("configs[2] hre-ques-im-hist + disc, fc7 4096, batch 20", dict(encoder='hre-ques-im-hist', decoder='disc', imgFeatureSize=4096, batchSize=20)),
    ("configs[3] mn-att-ques-im-hist + disc, 14x14x512, batch 20", dict(encoder='mn-att-ques-im-hist', decoder='disc', imgFeatureSize=512, imgSpatialSize=14, batchSize=20)),
    ("configs[4] mn-att-ques-im-hist + disc, 7x7x2048, bf16 option LSTM, batch 20", dict(encoder='mn-att-ques-im-hist', decoder='disc', imgFeatureSize=2048, imgSpatialSize=7, batchSize=20, lstmPrecision='bf16')),
]
hosts = sys.argv[1:] or ['native', 'python']
for name, kw in CONFIGS:
    for host in hosts:
        p = default_params(vocabSize=11322, maxHistoryLenPerRound=40, gpuid=0, **kw)
        model = NativeModel(p) if host == 'native' else Model(p)
        dl = SyntheticDataloader(p, seed=1, fast=True)
        for _ in range(4):
            model.trainIteration(dl)
        torch.cuda.synchronize()
        n = 12
        t0 = time.perf_counter()
        for _ in range(n):
            model.trainIteration(dl)
        if host == 'native':
            model.synchronize()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        print("%-80s %-7s %7.2f ms/step  %8.0f QA-rounds/s" % (name, host, dt * 1e3, p['batchSize'] * p['maxQuesCount'] / dt), flush=True)
        if host == 'native':
            model.close()
        del model
        torch.cuda.empty_cache()
