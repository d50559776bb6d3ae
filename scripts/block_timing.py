"""Per-workgroup phase breakdown of the option-LSTM forward step kernel (diagnostic build:
`make -C visdial_amd/csrc timing`).  Stamps are s_memtime cycles taken by wave 0 of each workgroup:
prologue (first tile in LDS) | K loop | epilogue transposes | epilogue loads+math+stores, and the
chip-wide residency profile of the launch (how many workgroups are alive over time: the tail)."""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from visdial_amd import _lib  # noqa: E402

_lib.LIB_PATH = os.path.join(ROOT, "visdial_amd", "libvisdial_hip_timing.so")
from visdial_amd import ops  # noqa: E402


def main():
    dev = "cuda"
    T, N, H, V = 4, int(os.environ.get("BT_N", 20000)), 512, 11322
    g = torch.Generator(device=dev).manual_seed(0)
    rnd = lambda *s: torch.randn(*s, device=dev, generator=g)
    Wh = rnd(H, 4 * H) * 0.04
    table = rnd(V + 1, 4 * H) * 0.1
    tok = torch.randint(0, V + 1, (T, N), device=dev, dtype=torch.int32, generator=g)
    gates = torch.empty(T, N, 4 * H, device=dev)
    h = torch.empty(T, N, H, device=dev)
    c = torch.empty(T, N, H, device=dev)
    for _ in range(3):
        ops.lstm_forward(table, Wh, gates, h, c, T, N, H, 0, 4 * H, tok_gather=tok)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    ops.lstm_forward(table, Wh, gates, h, c, T, N, H, 0, 4 * H, tok_gather=tok)
    e1.record()
    torch.cuda.synchronize()
    print("T=%d steps: %.3f ms (%.1f us / step)" % (T, e0.elapsed_time(e1), e0.elapsed_time(e1) / T * 1e3))
    lib = _lib.load()
    nb = ((N + 127) // 128) * 16
    buf = np.zeros(nb * 12, dtype=np.uint64)
    lib.vd_debug_timing.argtypes = [C.c_void_p, C.c_int]
    rc = lib.vd_debug_timing(buf.ctypes.data, buf.size)
    assert rc == 0
    t = buf.reshape(nb, 12).astype(np.int64)
    ok = (t[:, 0] > 0) & (t[:, 4] > t[:, 0])
    print('workgroups with complete stamps: %d of %d' % (ok.sum(), nb))
    t = t[ok]
    nb = len(t)
    pro, loop, tr, rest, bar = t[:, 1] - t[:, 0], t[:, 2] - t[:, 1], t[:, 3] - t[:, 2], t[:, 4] - t[:, 3], t[:, 5]
    tot = t[:, 4] - t[:, 0]
    span = t[:, 4].max() - t[:, 0].min()
    print("workgroups %d, kernel span %d cycles" % (nb, span))
    for name, v in (("prologue", pro), ("K loop", loop), ("epi transpose", tr),
                    ("epi loads/math/stores", rest), ("total", tot)):
        print("%-24s mean %8.0f  p10 %8.0f  p50 %8.0f  p90 %8.0f   %5.1f %% of total" % (
            name, v.mean(), np.percentile(v, 10), np.percentile(v, 50), np.percentile(v, 90), 100.0 * v.mean() / tot.mean()))
    # s_memtime bases differ between CUs; the residency profile uses s_memrealtime (100 MHz, chip-wide)
    r0, r1 = t[:, 6], t[:, 7]
    s0, s1 = r0.min(), r1.max()
    span = s1 - s0
    print("last launch: span %.1f us, mean workgroup lifetime %.1f us, mean resident %.0f of 768 slots" % (
        span / 100.0, (r1 - r0).mean() / 100.0, (r1 - r0).sum() / span))
    res = [int(((r0 <= s0 + f * span) & (r1 > s0 + f * span)).sum()) for f in np.linspace(0.025, 0.975, 20)]
    print("resident workgroups at 20 points across the launch:", res)
    if t[:, 8].min() > 0:
        l0, l1 = t[:, 8], t[:, 9]
        pts = np.linspace(0.02, 0.98, 49)
        inloop = [int(((l0 <= s0 + f * span) & (l1 > s0 + f * span)).sum()) for f in pts]
        alive = [int(((r0 <= s0 + f * span) & (r1 > s0 + f * span)).sum()) for f in pts]
        print("workgroups inside their K loop / alive at 49 points across the launch:")
        print(" ".join("%d/%d" % (a, b) for a, b in zip(inloop, alive)))
        print("time-averaged: %.0f in K loop, %.0f alive (768 slots); K-loop share of lifetime %.1f %%" % (
            (l1 - l0).sum() / span, (r1 - r0).sum() / span, 100.0 * (l1 - l0).sum() / (r1 - r0).sum()))
    print("shader clock during the launch ~ %.2f GHz" % (tot.mean() / ((r1 - r0).mean() * 10.0)))


if __name__ == "__main__":
    main()
