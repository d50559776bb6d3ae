"""The encoder's two-layer LSTM wavefront (vd_lstm2_forward / vd_lstm2_backward: T + 2 dependent grouped launches per direction) ALONE on an
idle chip: microseconds per tick at the shapes of BASELINE.json configs[1] (history: T up to 300 concatenated tokens, N = 200 rows, H = 512)
and of the headline's encoder (T = 40 / 20, N = 200), with all rows active and with a length-sorted ragged batch.

    python scripts/mb_ticks.py [T N H [bf16]]      (bf16: the ticks of a bf16 pass, csrc/lstm.hip vd_lstm2_forward_p / _backward_p)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from visdial_amd import ops  # noqa: E402


BF16 = 'bf16' in sys.argv[1:]
if BF16:
    sys.argv.remove('bf16')


def run(T, N, H, ragged, reps=3):
    g = torch.Generator(device='cuda').manual_seed(0)
    r = lambda *s: torch.randn(*s, device='cuda', generator=g)
    W = {k: r(H, 4 * H) * 0.04 for k in ('Wh1', 'Wx2', 'Wh2')}
    b2 = r(4 * H) * 0.1
    if ragged:
        lens = np.sort(np.random.RandomState(0).randint(T // 8, T + 1, size=N))[::-1]
        nact = np.array([(lens >= T - t).sum() for t in range(T)], np.int32)
    else:
        lens = np.full(N, T)
        nact = np.full(T, N, np.int32)
    tok = torch.zeros(T, N, dtype=torch.int32, device='cuda')
    for n in range(N):
        tok[T - int(lens[n]):, n] = 1
    st = dict(T=T, N=N, tok_mask=tok, b2=b2, gates1=r(T, N, 4 * H) * 0.3, h1=torch.zeros(T, N, H, device='cuda'),
              c1=torch.zeros(T, N, H, device='cuda'), gates2=torch.zeros(T, N, 4 * H, device='cuda'), h2=torch.zeros(T, N, H, device='cuda'),
              c2=torch.zeros(T, N, H, device='cuda'), nact=nact, **W)
    x1 = st['gates1'].clone()
    bw = dict(T=T, N=N, gates1=st['gates1'], c1=st['c1'], gates2=st['gates2'], c2=st['c2'], dh_last2=r(N, H) * 0.01,
              dh1_seq=torch.empty(T, N, H, device='cuda'), dc1=torch.empty(N, H, device='cuda'), dc2=torch.empty(N, H, device='cuda'), nact=nact, **W)
    out = []
    if BF16:
        fns = (('fwd', lambda: ops.lstm2_pass([st], H, ops.FLAG_BF16)), ('bwd', lambda: ops.lstm2_pass([bw], H, ops.FLAG_BF16, backward=True)))
    else:
        fns = (('fwd', lambda: ops.lstm2_forward([st], H)), ('bwd', lambda: ops.lstm2_backward([bw], H)))
    for name, fn in fns:
        ts = []
        for i in range(reps + 1):
            if name == 'fwd':
                st['gates1'].copy_(x1)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            fn()
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        ms = float(np.mean(ts[1:]))
        flop = 2.0 * float(nact.sum()) * 3 * H * 4 * H
        out.append("%s %7.3f ms = %5.1f us/tick (%d ticks), %5.1f TFLOP/s executed" % (name, ms, ms / (T + 2) * 1e3, T + 2, flop / ms / 1e9))
    print("%sT=%3d N=%3d H=%d %-7s mean active rows %5.1f | %s" % ('bf16 ticks ' if BF16 else '', T, N, H, 'ragged' if ragged else 'full', nact.mean(), ' | '.join(out)), flush=True)


if os.environ.get('MB_PHASES') == '1':
    pass
elif len(sys.argv) > 3:
    T, N, H = (int(x) for x in sys.argv[1:4])
    run(T, N, H, False)
    run(T, N, H, True)
else:
    for T, N in ((40, 200), (250, 200), (20, 200), (40, 64), (40, 32)):
        run(T, N, 512, False)
        run(T, N, 512, True)


def phases(label):
    """diagnostic library only (`make -C visdial_amd/csrc timing`, VD_LIB_PATH=.../libvisdial_hip_timing.so): where a workgroup of the LAST launch
    spent its time -- stamps of gemm_block (gemm_core.h): 0 start, 1 first tile staged, 2 K loop done, 3 epilogue transposes done, 4 end"""
    import ctypes as C
    from visdial_amd import _lib
    lib = _lib.load()
    if not hasattr(lib, 'vd_debug_timing'):
        return
    SL, NB = 12, 8192
    buf = (C.c_ulonglong * (SL * NB))()
    lib.vd_debug_timing(buf, SL * NB)
    a = np.frombuffer(buf, dtype=np.uint64).reshape(NB, SL).astype(np.int64)
    a = a[a[:, 6] > 0]
    a = a[a[:, 6] >= a[:, 6].max() - 3000]          # the last launch only: within 30 us (100 MHz clock)
    clk = (a[:, 4] - a[:, 0]) / np.maximum((a[:, 7] - a[:, 6]) / 100.0, 1e-3)            # shader cycles per us
    us = lambda x: np.median(x / clk)
    print("  %s last launch: %d workgroups; us (medians): fill %.1f | K loop %.1f | epilogue transposes %.1f | cell update + stores %.1f | total %.1f; "
          "launch span %.1f us; shader clock %.0f MHz" % (label, len(a), us(a[:, 1] - a[:, 0]), us(a[:, 2] - a[:, 1]), us(a[:, 3] - a[:, 2]),
                                                          us(a[:, 4] - a[:, 3]), np.median(a[:, 7] - a[:, 6]) / 100.0,
                                                          (a[:, 7].max() - a[:, 6].min()) / 100.0, np.median(clk)), flush=True)


if os.environ.get('MB_PHASES') == '1':
    for N in (200, 32):
        g = torch.Generator(device='cuda').manual_seed(0)
        T, H = 12, 512
        W = {k: torch.randn(H, 4 * H, device='cuda', generator=g) * 0.04 for k in ('Wh1', 'Wx2', 'Wh2')}
        st = dict(T=T, N=N, tok_mask=torch.ones(T, N, dtype=torch.int32, device='cuda'), b2=torch.zeros(4 * H, device='cuda'),
                  gates1=torch.randn(T, N, 4 * H, device='cuda', generator=g) * 0.3, h1=torch.zeros(T, N, H, device='cuda'), c1=torch.zeros(T, N, H, device='cuda'),
                  gates2=torch.zeros(T, N, 4 * H, device='cuda'), h2=torch.zeros(T, N, H, device='cuda'), c2=torch.zeros(T, N, H, device='cuda'),
                  nact=np.full(T, N, np.int32), **W)
        ops.lstm2_forward([st], H)
        torch.cuda.synchronize()
        phases('fwd N=%d (last tick = one L2 cell sub-problem)' % N)
        bw = dict(T=T, N=N, gates1=st['gates1'], c1=st['c1'], gates2=st['gates2'], c2=st['c2'], dh_last2=torch.randn(N, H, device='cuda', generator=g) * 0.01,
                  dh1_seq=torch.empty(T, N, H, device='cuda'), dc1=torch.empty(N, H, device='cuda'), dc2=torch.empty(N, H, device='cuda'), nact=st['nact'], **W)
        ops.lstm2_backward([bw], H)
        torch.cuda.synchronize()
        phases('bwd N=%d (last tick = one L1 cell sub-problem; stamp 3 is not set in the backward epilogue: read 2 -> 4 as one number)' % N)
