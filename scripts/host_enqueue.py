"""How long the HOST spends enqueueing a step (vd_model_forward_backward returns when everything is enqueued) against the device time of the
step, per BASELINE.json configuration.     python scripts/host_enqueue.py [configs ...]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault('GPU_MAX_HW_QUEUES', '1')
import torch  # noqa: E402,F401

from bench import config_params  # noqa: E402
from visdial_amd.dataloader import SyntheticDataloader  # noqa: E402
from visdial_amd.native import NativeModel  # noqa: E402
from visdial_amd._lib import call  # noqa: E402

for cfg in [int(a) for a in sys.argv[1:]] or [0, 1, 2, 3, 4]:
    p = config_params(cfg)
    model = NativeModel(p)
    dl = SyntheticDataloader(p, seed=1, fast=True)
    batch = dl.getTrainBatch(p)
    for _ in range(3):
        model.forwardBackward(batch)
        model.update()
    model.synchronize()
    enq, tot = [], []
    for _ in range(8):
        model.synchronize()
        t0 = time.perf_counter()
        call("vd_model_forward_backward", model.h, 0)
        t1 = time.perf_counter()
        model.synchronize()
        t2 = time.perf_counter()
        enq.append((t1 - t0) * 1e3)
        tot.append((t2 - t0) * 1e3)
        model.update()
    enq.sort(); tot.sort()
    print("configs[%d] %-22s host enqueue %.2f ms (median) of a %.2f ms forward + backward: %.0f %%" % (
        cfg, p['encoder'] + '+' + p['decoder'], enq[len(enq) // 2], tot[len(tot) // 2], 100 * enq[len(enq) // 2] / tot[len(tot) // 2]), flush=True)
    model.close()
