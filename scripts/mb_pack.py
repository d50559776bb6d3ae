"""Upper bound of what packing the option-LSTM backward steps together with the dWh GEMM could give:
run them sequentially on one stream vs concurrently on two (dependencies ignored)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from visdial_amd import ops

dev = "cuda"
T, N, H, V = 20, 20000, 512, 11322
g = torch.Generator(device=dev).manual_seed(0)
rnd = lambda *s: torch.randn(*s, device=dev, generator=g)
Wh = rnd(H, 4 * H) * 0.04
gates = torch.rand(T, N, 4 * H, device=dev, generator=g) * 0.5 + 0.25
gates2 = gates.clone()
h = rnd(T, N, H) * 0.1
c = rnd(T, N, H) * 0.1
dcw = torch.empty(N, H, device=dev)
dh_last = rnd(N, H)
dWh = torch.zeros(H, 4 * H, device=dev)
hh, gg = h.view(T * N, H), gates2.view(T * N, 4 * H)
K = (T - 1) * N
side = torch.cuda.Stream()
nchunk = int(os.environ.get("CHUNKS", 1))


def bwd():
    ops.lstm_backward(Wh, gates, c, dcw, T, N, H, dh_last=dh_last)


def wgrad():
    step = K // nchunk
    for i in range(nchunk):
        ops.gemm_tn_acc(hh[i * step:], gg[N + i * step:], dWh, M=H, N=4 * H, K=step)


def timed(fn):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1)


def seq():
    bwd()
    wgrad()


def conc():
    ev = torch.cuda.Event()
    ev.record()
    side.wait_event(ev)
    with torch.cuda.stream(side):
        wgrad()
    bwd()
    torch.cuda.current_stream().wait_stream(side)


for _ in range(2):
    seq(); conc()
print("sequential  bwd + dWh: %.2f ms" % min(timed(seq) for _ in range(3)))
print("concurrent  bwd | dWh (%d chunks): %.2f ms" % (nchunk, min(timed(conc) for _ in range(3))))
