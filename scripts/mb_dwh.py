"""The option recurrence's weight-gradient contraction dWh += h^T * da alone at the headline shape (M = 512, N = 2048, K = 19 x 20 000 rows):
fp32 MFMA (k-major LDS-DMA kernel) vs the exact split (csrc/split_core.h gemm_split_tn_kernel), ms and fp32-equivalent TFLOP/s, and the
error of both against an fp64 product on a K = 32 768 prefix.      python scripts/mb_dwh.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from visdial_amd import ops  # noqa: E402

M, N, K = 512, 2048, 19 * 20000
g = torch.Generator(device='cuda').manual_seed(0)
h = torch.tanh(torch.randn(K, M, device='cuda', generator=g))
da = torch.randn(K, N, device='cuda', generator=g) * 0.01
Ks = 32768
ref = (h[:Ks].double().t() @ da[:Ks].double())
for name, flags in (('fp32 MFMA', 0), ('split9', ops.FLAG_SPLIT9)):
    C = torch.zeros(M, N, device='cuda')
    ops.gemm_tn_acc(h[:Ks], da[:Ks], C, M=M, N=N, K=Ks, flags=flags)
    torch.cuda.synchronize()
    err = float((C.double() - ref).norm() / ref.norm())
    C = torch.zeros(M, N, device='cuda')
    ops.gemm_tn_acc(h, da, C, M=M, N=N, K=K, flags=flags)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        ops.gemm_tn_acc(h, da, C, M=M, N=N, K=K, flags=flags)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 3
    if flags == 0:
        full = C.clone()
    print("%-10s %7.3f ms  %6.1f TFLOP/s fp32-equivalent | rel-L2 vs fp64 (K = %d) %.2e | vs the fp32-MFMA result at full K %.2e" % (
        name, ms, 2.0 * M * N * K / ms / 1e9, Ks, err, float((C.double() / 4 - full.double() / 4).norm() / (full.double() / 4).norm())), flush=True)
