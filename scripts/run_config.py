"""Train one BASELINE.json configuration for a few steps (profiling target):  python scripts/run_config.py <config 1..4> [steps] [warmup]
Prints ms/step.  Fresh synthetic batch every step through the native (model-level) host, like bench.py."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault('GPU_MAX_HW_QUEUES', '1')
import torch  # noqa: E402

from bench import config_params  # noqa: E402
from visdial_amd.dataloader import SyntheticDataloader  # noqa: E402
from visdial_amd.native import NativeModel  # noqa: E402

cfg = int(sys.argv[1])
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
warm = int(sys.argv[3]) if len(sys.argv) > 3 else 2
p = config_params(cfg)
model = NativeModel(p)
dl = SyntheticDataloader(p, seed=1, fast=True)
for _ in range(warm):
    model.trainIteration(dl)
model.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    model.trainIteration(dl)
model.synchronize()
torch.cuda.synchronize()
print("configs[%d] %s + %s: %.3f ms/step" % (cfg, p['encoder'], p['decoder'], (time.perf_counter() - t0) / steps * 1e3))
model.close()
