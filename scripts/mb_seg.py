import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np, torch
from visdial_amd import ops
from microbench import timeit
dev = "cuda"
T, N, H, V = 20, 20000, 512, 11322
rng = np.random.RandomState(0)
ol = rng.randint(1, T + 1, size=N)
tok = np.zeros((N, T), np.int32)
for i, l in enumerate(ol):
    tok[i, :l] = rng.randint(1, V, size=l)
tokf = torch.from_numpy(np.ascontiguousarray(tok.T).reshape(-1)).to(dev)     # time-major, left-aligned options (pads ~50%)
X = torch.randn(T * N, 4 * H, device=dev)
offset = torch.empty(V + 2, dtype=torch.int32, device=dev); work = torch.empty(2 * (V + 1), dtype=torch.int32, device=dev)
perm = torch.empty(T * N, dtype=torch.int32, device=dev)
ops.token_sort(tokf, V + 1, offset, work, perm)
dtab = torch.zeros(V + 1, 4 * H, device=dev)
ms = timeit(lambda: ops.segment_rowsum_acc(X, tokf, perm, dtab), iters=5, warm=2)
print("segment rowsum (option-like tokens, 50%% pads) chunk=%s: %.3f ms  %.0f GB/s" % (os.environ.get('VD_SEG_CHUNK', '32'), ms, 4.0 * T * N * 4 * H / ms / 1e6))
