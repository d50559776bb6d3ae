"""The compact-state bf16 option recurrence (configs[4]) alone at the headline shape: ms per direction, per-launch time and the HBM rate of
its algorithmic bytes.  With the diagnostic library (`make -C visdial_amd/csrc timing`, VD_LIB_PATH=visdial_amd/libvisdial_hip_timing.so)
it also prints where a workgroup of the LAST launch spent its time (pipeline fill | K loop | epilogue, 100 MHz chip clock).
    python scripts/mb_c16.py [T N H]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from visdial_amd import _lib  # noqa: E402

T, N, H = (int(x) for x in sys.argv[1:4]) if len(sys.argv) > 3 else (20, 20000, 512)
V = 11322
lib = _lib.load()
fwd = getattr(lib, '_Z19vd_lstm_forward_c16PKtlPKiPKfPtS5_PfS6_iiiP12ihipStream_t')
bwd = getattr(lib, '_Z20vd_lstm_backward_c16PKfPtS0_S0_PfiiiP12ihipStream_t')
p = C.c_void_p
fwd.argtypes = [p, C.c_long, p, p, p, p, p, p, C.c_int, C.c_int, C.c_int, p]
bwd.argtypes = [p, p, p, p, p, C.c_int, C.c_int, C.c_int, p]
g = torch.Generator(device='cuda').manual_seed(0)
Wh = torch.randn(H, 4 * H, device='cuda', generator=g) * 0.04
tab16 = (torch.randn(V + 1, 4 * H, device='cuda', generator=g) * 0.5).to(torch.bfloat16)
tok = torch.randint(1, V + 1, (T, N), device='cuda', generator=g, dtype=torch.int32)
gates16 = torch.empty(T, N, 4 * H, device='cuda', dtype=torch.bfloat16)
h16 = torch.empty(T, N, H, device='cuda', dtype=torch.bfloat16)
h_last = torch.empty(N, H, device='cuda')
c = torch.empty(T, N, H, device='cuda')
dc = torch.empty(N, H, device='cuda')
dh_last = torch.randn(N, H, device='cuda', generator=g) * 0.01
stream = torch.cuda.current_stream().cuda_stream


def run_f():
    rc = fwd(tab16.data_ptr(), 4 * H, tok.data_ptr(), Wh.data_ptr(), gates16.data_ptr(), h16.data_ptr(), h_last.data_ptr(), c.data_ptr(), T, N, H, stream)
    assert rc == 0, lib.vd_last_error()


def run_b():
    rc = bwd(Wh.data_ptr(), gates16.data_ptr(), c.data_ptr(), dh_last.data_ptr(), dc.data_ptr(), T, N, H, stream)
    assert rc == 0, lib.vd_last_error()


def phases(label):
    if not hasattr(lib, 'vd_debug_timing'):
        return
    SL, NB = 12, 8192
    buf = (C.c_ulonglong * (SL * NB))()
    lib.vd_debug_timing(buf, SL * NB)
    a = np.frombuffer(buf, dtype=np.uint64).reshape(NB, SL).astype(np.int64)
    a = a[a[:, 6] > 0]
    a = a[a[:, 6] >= a[:, 6].max() - 100000]          # the last launch only (stale rows of earlier, larger grids dropped): within 1 ms
    t0 = a[:, 6].min()
    fill, kloop, epi = (a[:, 8] - a[:, 6]) / 100.0, (a[:, 9] - a[:, 8]) / 100.0, (a[:, 7] - a[:, 9]) / 100.0
    if a[:, 3].max() > 0 and (a[:, 3] > a[:, 0]).all():        # the wave-asynchronous kernel stamps wave 0's first block
        print("  %s wave 0, first block, us: operand requests %.1f | K loop %.1f | epilogue %.1f | second block ends at %.1f after the first began" % (
            label, np.median(a[:, 1] - a[:, 0]) / 100.0, np.median(a[:, 2] - a[:, 1]) / 100.0, np.median(a[:, 3] - a[:, 2]) / 100.0,
            np.median(a[:, 4] - a[:, 0]) / 100.0), flush=True)
    print("  %s last launch, %d workgroups, us: fill %.1f | K loop %.1f | epilogue %.1f (medians); launch span %.1f; starts p50 %.1f p90 %.1f" % (
        label, len(a), np.median(fill), np.median(kloop), np.median(epi), (a[:, 7].max() - t0) / 100.0,
        np.median(a[:, 6] - t0) / 100.0, np.percentile(a[:, 6] - t0, 90) / 100.0), flush=True)


# algorithmic bytes per timestep launch: forward = table row gather (bf16) + gates16 write + h16 read/write + c read/write;
# backward = gates16[t] read + da16[t+1] read + da16[t] write (in place) + c[t], c[t-1] reads + dc read/write
bytes_f = N * 4 * H * 2 * 2 + N * H * 2 * 2 + N * H * 4 * 2
bytes_b = N * 4 * H * 2 * 3 + N * H * 4 * 2 + N * H * 4 * 2
for name, fn, nbytes in (('fwd', run_f, bytes_f), ('bwd', run_b, bytes_b)):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        fn()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 3
    us = ms / (T - 1) * 1e3
    print("c16 %s %7.3f ms  %6.1f us/launch  %5.0f MB/launch algorithmic -> %5.2f TB/s" % (name, ms, us, nbytes / 1e6, nbytes / us / 1e6), flush=True)
    phases(name)
