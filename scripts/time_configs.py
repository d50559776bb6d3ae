"""Step time of the other BASELINE.json configurations (parity-test cases; informative only)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from visdial_amd.opts import default_params
from visdial_amd.dataloader import SyntheticDataloader
from visdial_amd.model import Model

CONFIGS = [
    ("configs[1] lf-ques-im-hist + gen, fc7 4096, batch 20", dict(encoder='lf-ques-im-hist', decoder='gen', imgFeatureSize=4096, batchSize=20)),
    ("configs[2] hre-ques-im-hist + disc, fc7 4096, batch 20", dict(encoder='hre-ques-im-hist', decoder='disc', imgFeatureSize=4096, batchSize=20)),
    ("configs[3] mn-att-ques-im-hist + disc, 14x14x512, batch 20", dict(encoder='mn-att-ques-im-hist', decoder='disc', imgFeatureSize=512, imgSpatialSize=14, batchSize=20)),
    ("configs[4] mn-att-ques-im-hist + disc, 7x7x2048, bf16 option LSTM, batch 20", dict(encoder='mn-att-ques-im-hist', decoder='disc', imgFeatureSize=2048, imgSpatialSize=7, batchSize=20, lstmPrecision='bf16')),
]
for name, kw in CONFIGS:
    p = default_params(vocabSize=11322, maxHistoryLenPerRound=40, gpuid=0, **kw)
    model = Model(p)
    dl = SyntheticDataloader(p, seed=1)
    batch = dl.getTrainBatch(p)
    prepared = model.prepare_inputs(batch)

    def step():
        model.wrapper.zeroGradParameters()
        loss = model.forwardBackward(batch, prepared=prepared)
        model.update()
        return loss
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 10
    for _ in range(n):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print("%-78s %7.2f ms/step  %8.0f QA-rounds/s" % (name, dt * 1e3, p['batchSize'] * p['maxQuesCount'] / dt))
    del model
    torch.cuda.empty_cache()
