"""The encoder's two-layer LSTM wavefront (vd_lstm2_forward / vd_lstm2_backward: T + 2 dependent grouped launches per direction) ALONE on an
idle chip: microseconds per tick at the shapes of BASELINE.json configs[1] (history: T up to 300 concatenated tokens, N = 200 rows, H = 512)
and of the headline's encoder (T = 40 / 20, N = 200), with all rows active and with a length-sorted ragged batch.

    python scripts/mb_ticks.py [T N H]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from visdial_amd import ops  # noqa: E402


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
    for name, fn in (('fwd', lambda: ops.lstm2_forward([st], H)), ('bwd', lambda: ops.lstm2_backward([bw], H))):
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
    print("T=%3d N=%3d H=%d %-7s mean active rows %5.1f | %s" % (T, N, H, 'ragged' if ragged else 'full', nact.mean(), ' | '.join(out)), flush=True)


if len(sys.argv) > 3:
    T, N, H = (int(x) for x in sys.argv[1:4])
    run(T, N, H, False)
    run(T, N, H, True)
else:
    for T, N in ((40, 200), (250, 200), (20, 200), (40, 64), (40, 32)):
        run(T, N, 512, False)
        run(T, N, 512, True)
