"""The embedding-sized products of the projection-table gradient alone (E = 300, V + 1 = 11 323, 4H = 2048):
dEmb += dTable * Wx^T (vd_gemm_nt, atomic accumulation), dWx += Emb^T * dTable (vd_gemm_tn_acc), table = Emb * Wx + b (vd_gemm_nn).
    python scripts/mb_embed_gemms.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from visdial_amd import ops  # noqa: E402
from visdial_amd._lib import call  # noqa: E402

V1, E, H4 = 11323, 300, 2048
g = torch.Generator(device='cuda').manual_seed(0)
emb = torch.randn(V1, E, device='cuda', generator=g) * 0.1
wx = torch.randn(E, H4, device='cuda', generator=g) * 0.05
dtab = torch.randn(V1, H4, device='cuda', generator=g) * 0.01
bias = torch.randn(H4, device='cuda', generator=g) * 0.1


def timed(fn, n=5):
    ts = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return sorted(ts)[n // 2] * 1e3


flop = 2.0 * V1 * E * H4
demb = torch.zeros(V1, E, device='cuda')
stream = lambda: torch.cuda.current_stream().cuda_stream
f1 = lambda: call("vd_gemm_nt", dtab.data_ptr(), H4, wx.data_ptr(), H4, None, demb.data_ptr(), E, V1, E, H4, 0, 2, stream())
f1()
torch.cuda.synchronize()
err1 = float((demb.double() - dtab.double() @ wx.double().t()).norm() / (dtab.double() @ wx.double().t()).norm())
us = timed(f1)
print("dEmb += dTable * Wx^T   %7.1f us  %6.1f TFLOP/s  rel-L2 of the first call %.1e" % (us, flop / us / 1e6, err1))
dwx = torch.zeros(E, H4, device='cuda')
f2 = lambda: ops.gemm_tn_acc(emb, dtab, dwx, M=E, N=H4, K=V1)
f2()
torch.cuda.synchronize()
err2 = float((dwx.double() - emb.double().t() @ dtab.double()).norm() / (emb.double().t() @ dtab.double()).norm())
us = timed(f2)
print("dWx  += Emb^T * dTable  %7.1f us  %6.1f TFLOP/s  rel-L2 of the first call %.1e" % (us, flop / us / 1e6, err2))
tab = torch.empty(V1, H4, device='cuda')
f3 = lambda: ops.gemm_nn(emb, wx, tab, bias=bias)
f3()
torch.cuda.synchronize()
ref = emb.double() @ wx.double() + bias.double()
err3 = float((tab.double() - ref).norm() / ref.norm())
us = timed(f3)
print("table = Emb * Wx + b    %7.1f us  %6.1f TFLOP/s  rel-L2 %.1e" % (us, flop / us / 1e6, err3))
