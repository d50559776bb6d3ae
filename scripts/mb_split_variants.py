"""A/B of the split9 recurrence kernels at the headline shape (T = 20, N = 20 000, H = 512) on a VARIANT build of the library
(`make -C visdial_amd/csrc variant NAME=sv DEFS=-DVD_SPLIT_VARIANTS`, loaded through VD_LIB_PATH): the kernel is picked per call by
the environment variables VD_SPLIT_FWD / VD_SPLIT_BWD (csrc/lstm.hip).  Prints us / launch and the deviation of every variant's results
from variant 0 (round 4's kernel).  Never part of the product.

    VD_LIB_PATH=visdial_amd/libvisdial_hip_sv.so python scripts/mb_split_variants.py [fwd ids] [bwd ids]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from visdial_amd import ops  # noqa: E402

T, N, H = 20, 20000, 512
V = 11322
FWD = [int(x) for x in sys.argv[1].split(',')] if len(sys.argv) > 1 and sys.argv[1] else []
BWD = [int(x) for x in sys.argv[2].split(',')] if len(sys.argv) > 2 and sys.argv[2] else []
REPS = int(os.environ.get('MB_REPS', '3'))
ZERO = os.environ.get('MB_ZERO') == '1'     # all-zero operands: the matrix pipe draws less power and the clock stays up (DVFS) -- separates power from structure
g = torch.Generator(device='cuda').manual_seed(0)
Wh = torch.randn(H, 4 * H, device='cuda', generator=g) * 0.04
tab = torch.randn(V + 1, 4 * H, device='cuda', generator=g) * 0.5
if ZERO:
    Wh.zero_()
    tab.zero_()
tok = torch.randint(1, V + 1, (T, N), device='cuda', generator=g, dtype=torch.int32)
gates = torch.empty(T, N, 4 * H, device='cuda')
h = torch.empty(T, N, H, device='cuda')
c = torch.empty(T, N, H, device='cuda')
dc = torch.empty(N, H, device='cuda')
dh_last = torch.randn(N, H, device='cuda', generator=g) * 0.01
flop = 2.0 * N * H * 4 * H * (T - 1)


def timed(fn):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(REPS):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / REPS


def rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30))


def fwd(flags=ops.FLAG_SPLIT9):
    ops.lstm_forward(tab, Wh, gates, h, c, T, N, H, 0, 4 * H, tok_gather=tok, flags=flags)


def bwd(flags=ops.FLAG_SPLIT9):
    ops.lstm_backward(Wh, gates, c, dc, T, N, H, dh_last=dh_last, flags=flags)


os.environ['VD_SPLIT_FWD'] = '0'
os.environ['VD_SPLIT_BWD'] = '0'
ms = timed(lambda: fwd(0))
print("fwd fp32      %7.3f ms  %6.1f us/launch" % (ms, ms / (T - 1) * 1e3), flush=True)
fwd()
torch.cuda.synchronize()
h_ref, g_ref = h[-1].clone(), gates[-1].clone()
for v in [0] + FWD:
    os.environ['VD_SPLIT_FWD'] = str(v)
    ms = timed(fwd)
    torch.cuda.synchronize()
    dev = "K loop only" if v >= 10 else "h %.1e gates %.1e vs v0" % (rel(h[-1], h_ref), rel(gates[-1], g_ref))
    print("fwd variant %2d %7.3f ms  %6.1f us/launch  %6.1f TF fp32-equiv  (%s)" % (v, ms, ms / (T - 1) * 1e3, flop / ms / 1e9, dev), flush=True)
os.environ['VD_SPLIT_FWD'] = '0'
fwd()
gates0 = gates.clone()
ms0 = None
for v in [-1, 0] + BWD:
    os.environ['VD_SPLIT_BWD'] = str(max(v, 0))
    times = []
    for _ in range(REPS + 1):          # the backward pass overwrites the saved gates: restore them outside the timed region
        gates.copy_(gates0)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        bwd(0 if v < 0 else ops.FLAG_SPLIT9)
        e1.record()
        torch.cuda.synchronize()
        times.append(e0.elapsed_time(e1))
    ms = sum(times[1:]) / REPS
    if v == 0:
        da_ref, dc_ref = gates[0].clone(), dc.clone()
    dev = "fp32 kernels" if v < 0 else "da0 %.1e dc0 %.1e vs v0" % (rel(gates[0], da_ref), rel(dc, dc_ref))
    print("bwd variant %2d %7.3f ms  %6.1f us/launch  %6.1f TF fp32-equiv  (%s)" % (v, ms, ms / (T - 1) * 1e3, flop / ms / 1e9, dev), flush=True)
