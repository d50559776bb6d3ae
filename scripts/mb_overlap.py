"""How much does a concurrent latency-shape recurrence (encoder-sized LSTM on a side stream) slow the
throughput-shape option-LSTM forward on the main stream?"""
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
Ns, Ts = 200, 40
toks = torch.randint(1, V, (Ts, Ns), device=dev, dtype=torch.int32, generator=g)
xp = rnd(Ts, Ns, 4 * H)
gs = torch.empty(Ts, Ns, 4 * H, device=dev)
hs = torch.empty(Ts, Ns, H, device=dev)
cs = torch.empty(Ts, Ns, H, device=dev)
side = torch.cuda.Stream(priority=-1)


def big():
    ops.lstm_forward(table, Wh, gates, h, c, T, N, H, 0, 4 * H, tok_gather=tok)


def small(reps):
    for _ in range(reps):
        ops.lstm_forward(xp, Wh, gs, hs, cs, Ts, Ns, H, Ns * 4 * H, 4 * H, tok_mask=toks)


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
            small(reps)
            s1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1), (s0.elapsed_time(s1) if reps else 0.0)


for _ in range(2):
    run(2)
for reps in (0, 1, 2, 4, 8):
    r = [run(reps) for _ in range(3)]
    print("side recurrences x%d: option fwd %.2f ms, side chain %.2f ms (%.1f us/step)" % (
        reps, min(a for a, _ in r), min(b for _, b in r), min(b for _, b in r) * 1e3 / max(1, reps * Ts)))
