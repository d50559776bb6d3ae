"""Target of the rocprofv3 --pmc passes (profiles/r02_pmc_*.txt): the three option-LSTM kernel families at the headline
shape, default configuration, a few launches each -- nothing else, so per-kernel counter averages are clean."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from visdial_amd import ops

T, N, H, V = 20, 20000, 512, 11322
FLAGS = ops.PRECISION_FLAGS[sys.argv[1]] if len(sys.argv) > 1 else 0        # fp32 (default) | bf16 | split9 | split6
g = torch.Generator(device="cuda").manual_seed(0)
rnd = lambda *s: torch.randn(*s, device="cuda", generator=g)
Wh, table = rnd(H, 4 * H) * 0.04, rnd(V + 1, 4 * H) * 0.1
tok = torch.randint(0, V + 1, (T, N), device="cuda", dtype=torch.int32, generator=g)
gates, h, c = torch.empty(T, N, 4 * H, device="cuda"), torch.empty(T, N, H, device="cuda"), torch.empty(T, N, H, device="cuda")
dcw, dh_last, dWh = torch.empty(N, H, device="cuda"), rnd(N, H), torch.zeros(H, 4 * H, device="cuda")
for _ in range(3):
    ops.lstm_forward(table, Wh, gates, h, c, T, N, H, 0, 4 * H, tok_gather=tok, flags=FLAGS)
    ops.lstm_backward(Wh, gates, c, dcw, T, N, H, dh_last=dh_last, flags=FLAGS)
    ops.gemm_tn_acc(h.view(T * N, H), gates.view(T * N, 4 * H)[N:], dWh, M=H, N=4 * H, K=(T - 1) * N, flags=FLAGS & 3)
torch.cuda.synchronize()
print("pmc target done")
