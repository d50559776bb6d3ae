"""The option recurrence alone at the headline shape (T = 20, N = 20 000, H = 512), per arithmetic: ms per direction and the
fp32-equivalent TFLOP/s of the recurrent products.   python scripts/mb_recurrence.py [T N H [fp32,bf16,...]]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from visdial_amd import ops  # noqa: E402

T, N, H = (int(x) for x in sys.argv[1:4]) if len(sys.argv) > 3 else (20, 20000, 512)
V = 11322
g = torch.Generator(device='cuda').manual_seed(0)
Wh = torch.randn(H, 4 * H, device='cuda', generator=g) * 0.04
tab = torch.randn(V + 1, 4 * H, device='cuda', generator=g) * 0.5
tok = torch.randint(1, V + 1, (T, N), device='cuda', generator=g, dtype=torch.int32)
gates = torch.empty(T, N, 4 * H, device='cuda')
h = torch.empty(T, N, H, device='cuda')
c = torch.empty(T, N, H, device='cuda')
dc = torch.empty(N, H, device='cuda')
dh_last = torch.randn(N, H, device='cuda', generator=g) * 0.01
flop = 2.0 * N * H * 4 * H * (T - 1)
ONLY = sys.argv[4].split(',') if len(sys.argv) > 4 else None
for name, flags in (('fp32', 0), ('split9', ops.FLAG_SPLIT9), ('split6', ops.FLAG_SPLIT6), ('split3', ops.FLAG_SPLIT3), ('bf16', ops.FLAG_BF16)):
    if ONLY and name not in ONLY:
        continue
    res = []
    for fn in (lambda: ops.lstm_forward(tab, Wh, gates, h, c, T, N, H, 0, 4 * H, tok_gather=tok, flags=flags),
               lambda: ops.lstm_backward(Wh, gates, c, dc, T, N, H, dh_last=dh_last, flags=flags)):
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            fn()
        e1.record()
        torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) / 3)
    print("%-7s fwd %7.3f ms (%6.1f us/launch, %6.1f TF fp32-equiv) | bwd %7.3f ms (%6.1f us/launch, %6.1f TF fp32-equiv)" % (
        name, res[0], res[0] / (T - 1) * 1e3, flop / res[0] / 1e9, res[1], res[1] / (T - 1) * 1e3, flop / res[1] / 1e9), flush=True)
