"""A weight-gradient contraction of the encoder's recurrences alone (dW += X^T * dA: M = H = 512, N = 4H = 2048, K = T * N rows of fp32
state): fp32 MFMA (k-major LDS-DMA kernel) vs the bf16 pass's arithmetic (fp32 rows rounded to bf16 in registers, gemm_split_tn_kernel<1>).
    python scripts/mb_enc_wgrad.py [K]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from visdial_amd import ops  # noqa: E402

M, N = 512, 2048
g = torch.Generator(device='cuda').manual_seed(0)
for K in ([int(sys.argv[1])] if len(sys.argv) > 1 else [8000, 4000, 7800]):
    h = torch.tanh(torch.randn(K, M, device='cuda', generator=g))
    da = torch.randn(K, N, device='cuda', generator=g) * 0.01
    ref = h.double().t() @ da.double()
    for name, flags in (('fp32 MFMA', 0), ('bf16', ops.FLAG_BF16)):
        C = torch.zeros(M, N, device='cuda')
        ops.gemm_tn_acc(h, da, C, M=M, N=N, K=K, flags=flags)
        torch.cuda.synchronize()
        err = float((C.double() - ref).norm() / ref.norm())
        ts = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            ops.gemm_tn_acc(h, da, C, M=M, N=N, K=K, flags=flags)
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        ms = sorted(ts)[2]
        print("K=%5d %-10s %7.1f us  %6.1f TFLOP/s | rel-L2 vs fp64 %.2e" % (K, name, ms * 1e3, 2.0 * M * N * K / ms / 1e9, err), flush=True)
