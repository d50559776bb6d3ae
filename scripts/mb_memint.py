"""Probe: is the option-LSTM forward K loop (LDS-DMA prefetch depth 2 for A, 1 for B) sensitive to memory traffic of
OTHER work on the chip?  Main stream: the 20-step option recurrence, with its full epilogue and as the K-loop-only
diagnostic (VD_LSTM_FWD_EPI_SEQ=2).  Side stream: a streaming elementwise kernel (axpby, 12 B/element) looping for the
whole duration."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from visdial_amd import ops

dev = "cuda"
T, N, H, V = 20, 20000, 512, 11322
g = torch.Generator(device=dev).manual_seed(0)
rnd = lambda *s: torch.randn(*s, device=dev, generator=g)
Wh = rnd(H, 4 * H) * 0.04
table = rnd(V + 1, 4 * H) * 0.1
tok = torch.randint(0, V + 1, (T, N), device=dev, dtype=torch.int32, generator=g)
gates = torch.empty(T, N, 4 * H, device=dev)
h = torch.empty(T, N, H, device=dev)
c = torch.empty(T, N, H, device=dev)
n_el = int(os.environ.get("MB_SIDE_ELEMS", 32 << 20))
xa, xb, xc = rnd(n_el), rnd(n_el), torch.empty(n_el, device=dev)
side = torch.cuda.Stream(priority=-1)
fl = 2.0 * N * H * 4 * H * (T - 1)


def big():
    ops.lstm_forward(table, Wh, gates, h, c, T, N, H, 0, 4 * H, tok_gather=tok)


def run(reps):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s0, s1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    big()
    e1.record()
    if reps:
        side.wait_event(e0)
        with torch.cuda.stream(side):
            s0.record()
            for _ in range(reps):
                ops.axpby(xa, xb, xc, 1.0, 1.0)
            s1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1), (s0.elapsed_time(s1) if reps else 0.0)


for name, knobs in (("full epilogue", {}), ("K loop only", dict(VD_LSTM_FWD_EPI_SEQ=2))):
    ops.tune_clear()
    for k, v in knobs.items():
        ops.tune_set(k, v)
    for _ in range(2):
        run(4)
    for reps in (0, 8, 16, 32, 64):
        r = [run(reps) for _ in range(3)]
        tb, ts = min(a for a, _ in r), min(b for _, b in r)
        print("[%-13s] side axpby x%-3d: option fwd %.2f ms = %.1f TF | side %.2f ms = %.2f TB/s" % (
            name, reps, tb, fl / tb / 1e9, ts, (reps * 12.0 * n_el / (ts * 1e-3) / 1e12) if reps else 0.0), flush=True)
ops.tune_clear()
