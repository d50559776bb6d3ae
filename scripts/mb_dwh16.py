"""The bf16 pass's two consumers of the saved gate gradients (configs[4]), ALONE at the headline shape: the weight-gradient contraction
dWh += h16^T * da16 (vd_gemm_tn_acc_bf16: M = H, N = 4H, K = (T - 1) * N rows) and the projection-table gradient's row sums over the token
order (vd_segment_rowsum_acc_bf16).  Prints ms per launch, the bf16 MFMA rate of the contraction and the HBM rate of the bytes both read.
    python scripts/mb_dwh16.py [T N H]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from visdial_amd import _lib  # noqa: E402

T, N, H = (int(x) for x in sys.argv[1:4]) if len(sys.argv) > 3 else (20, 20000, 512)
lib = _lib.load()
tn = getattr(lib, '_Z19vd_gemm_tn_acc_bf16PKtS0_PfliiiP12ihipStream_t')
p = C.c_void_p
tn.argtypes = [p, p, p, C.c_long, C.c_int, C.c_int, C.c_int, p]
g = torch.Generator(device='cuda').manual_seed(0)
K = (T - 1) * N
h16 = (torch.randn(K, H, device='cuda', generator=g) * 0.3).to(torch.bfloat16)
da16 = (torch.randn(K, 4 * H, device='cuda', generator=g) * 0.01).to(torch.bfloat16)
dW = torch.zeros(H, 4 * H, device='cuda')
stream = torch.cuda.current_stream().cuda_stream


def run():
    rc = tn(h16.data_ptr(), da16.data_ptr(), dW.data_ptr(), 4 * H, H, 4 * H, K, stream)
    assert rc == 0, lib.vd_last_error()


run()
torch.cuda.synchronize()
ref = h16[:16384].float().t() @ da16[:16384].float()
dW.zero_()
rc = tn(h16.data_ptr(), da16.data_ptr(), dW.data_ptr(), 4 * H, H, 4 * H, 16384, stream)
torch.cuda.synchronize()
print("check (K = 16384): max |diff| %.3e of max |ref| %.3e" % ((dW - ref).abs().max().item(), ref.abs().max().item()))
ts = []
for i in range(6):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    run()
    e1.record()
    torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1))
ms = sorted(ts[1:])[len(ts[1:]) // 2]
flops = 2.0 * H * 4 * H * K
nbytes = K * (H + 4 * H) * 2
print("dWh16 M=%d N=%d K=%d: %.3f ms  = %.0f TFLOP/s bf16, %.2f TB/s of its %.2f GB" % (H, 4 * H, K, ms, flops / ms / 1e9, nbytes / ms / 1e9, nbytes / 1e9))

# the projection-table gradient's row sums over the same da16 rows + those of step 0 (T * N rows, token order)
rs = getattr(lib, '_Z26vd_segment_rowsum_acc_bf16PKtlPKiS2_liPflP12ihipStream_t')
rs.argtypes = [p, C.c_long, p, p, C.c_long, C.c_int, p, C.c_long, p]
V1 = 11323
n = T * N
da_all = (torch.randn(n, 4 * H, device='cuda', generator=g) * 0.01).to(torch.bfloat16)
tok = torch.randint(1, V1, (n,), device='cuda', generator=g, dtype=torch.int32)
perm = torch.argsort(tok.long(), stable=True).to(torch.int32)
dtab = torch.zeros(V1, 4 * H, device='cuda')
ts = []
for i in range(6):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    assert rs(da_all.data_ptr(), 4 * H, tok.data_ptr(), perm.data_ptr(), n, 4 * H, dtab.data_ptr(), 4 * H, stream) == 0
    e1.record()
    torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1))
ms = sorted(ts[1:])[len(ts[1:]) // 2]
print("segment row sums [%d x %d] bf16 -> [%d x %d]: %.3f ms = %.2f TB/s of its %.2f GB" % (n, 4 * H, V1, 4 * H, ms, n * 4 * H * 2 / ms / 1e9, n * 4 * H * 2 / 1e9))
