"""Kernel micro-benchmarks at the headline shapes (mn-att-ques-im-hist + disc, B=20):
prints achieved TFLOP/s (fp32 MFMA peak 157.3) or GB/s per kernel family."""
import sys
import os

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from visdial_amd import ops


def timeit(fn, iters=5, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters  # ms


def main():
    only_big = len(sys.argv) > 1 and sys.argv[1] == "big"
    dev = "cuda"
    T, N, H, E, V = 20, int(os.environ.get("MB_N", 20000)), 512, 300, 11322
    g = torch.Generator(device=dev).manual_seed(0)
    rnd = lambda *s: torch.randn(*s, device=dev, generator=g)
    Wh = rnd(H, 4 * H) * 0.04
    table = rnd(V + 1, 4 * H) * 0.1
    tok = torch.randint(0, V + 1, (T, N), device=dev, dtype=torch.int32, generator=g)
    gates = torch.empty(T, N, 4 * H, device=dev)
    h = torch.empty(T, N, H, device=dev)
    c = torch.empty(T, N, H, device=dev)
    dcw = torch.empty(N, H, device=dev)
    dh_last = rnd(N, H)

    fl = 2.0 * N * H * 4 * H * (T - 1)
    dWh = torch.zeros(H, 4 * H, device=dev)
    hh = h.view(T * N, H)
    gg = gates.view(T * N, 4 * H)
    K = (T - 1) * N

    def bwd():
        ops.lstm_backward(Wh, gates, c, dcw, T, N, H, dh_last=dh_last)

    def fwd():
        ops.lstm_forward(table, Wh, gates, h, c, T, N, H, 0, 4 * H, tok_gather=tok)

    # A/B sweep of the recurrence launch structure (one process, knobs through vd_tune_set)
    variants = [("per-step, bwd epilogue batch 1 (<=128 VGPR)", dict(VD_LSTM_BWD_BATCH2=0)),
                ("per-step launches", dict()),
                ("per-step, fwd epilogue compiler-scheduled (r1)", dict(VD_LSTM_FWD_EPI_SEQ=1)),
                ("per-step, fwd K LOOP ONLY (diagnostic, no epilogue)", dict(VD_LSTM_FWD_EPI_SEQ=2)),
                ("per-step launches (repeat)", dict()),
                ("per-step, bwd batch 2 (<=168 VGPR)", dict(VD_LSTM_BWD_BATCH2=1)),
                ("persistent", dict(VD_LSTM_PERSIST_FWD=1, VD_LSTM_PERSIST_BWD=1)),
                ("persistent, bwd batch 1", dict(VD_LSTM_PERSIST_FWD=1, VD_LSTM_PERSIST_BWD=1, VD_LSTM_BWD_BATCH2=0)),
                ("persistent, stagger 40us", dict(VD_LSTM_PERSIST_FWD=1, VD_LSTM_PERSIST_BWD=1, VD_LSTM_STAGGER_US=40)),
                ("persistent, 2 WG/CU grid", dict(VD_LSTM_PERSIST_FWD=1, VD_LSTM_PERSIST_BWD=1, VD_LSTM_SEQ_WGS_PER_CU=2))]
    if os.environ.get("MB_SWEEP", "1") == "0":
        variants = variants[:6]
    for name, knobs in variants:
        ops.tune_clear()
        for k, v in knobs.items():
            ops.tune_set(k, v)
        ms = timeit(fwd, iters=5, warm=2)
        ms2 = timeit(bwd, iters=3, warm=1)
        bad = ops.lstm_seq_status()
        print("option LSTM T=20 N=%d [%-44s] fwd %.2f ms %.1f TF | bwd %.2f ms %.1f TF%s" % (
            N, name, ms, fl / ms / 1e9, ms2, fl / ms2 / 1e9, "  TIMEOUT FLAG SET" if bad else ""))
    ops.tune_clear()
    fwd()
    bwd()
    for name, knobs in [("1024 blocks (round 1)", dict()), ("768 blocks", dict(VD_TN_BLOCKS=768)),
                        ("512 blocks", dict(VD_TN_BLOCKS=512)), ("1536 blocks", dict(VD_TN_BLOCKS=1536)),
                        ("768 blocks, no rotation", dict(VD_TN_BLOCKS=768, VD_GEMM_ROTATE=0)),
                        ("1024 blocks, no rotation", dict(VD_GEMM_ROTATE=0)),
                        ("768 blocks, double-buffered cfg", dict(VD_TN_BLOCKS=768, VD_TN_CFG=0)),
                        ("512 blocks, double-buffered cfg", dict(VD_TN_BLOCKS=512, VD_TN_CFG=0)),
                        ("768 blocks, k-major LDS-DMA", dict(VD_TN_BLOCKS=768, VD_TN_CFG=20)),
                        ("512 blocks, k-major LDS-DMA", dict(VD_TN_BLOCKS=512, VD_TN_CFG=20)),
                        ("1024 blocks, k-major LDS-DMA", dict(VD_TN_BLOCKS=1024, VD_TN_CFG=20)),
                        ("1536 blocks, k-major LDS-DMA", dict(VD_TN_BLOCKS=1536, VD_TN_CFG=20)),
                        ("768 blocks, k-major, no rot", dict(VD_TN_BLOCKS=768, VD_TN_CFG=20, VD_GEMM_ROTATE=0)),
                        ("2304 blocks, k-major", dict(VD_TN_BLOCKS=2304, VD_TN_CFG=20))]:
        ops.tune_clear()
        for k, v in knobs.items():
            ops.tune_set(k, v)
        ms = timeit(lambda: ops.gemm_tn_acc(hh, gg[N:], dWh, M=H, N=4 * H, K=K), iters=3, warm=1)
        print("dWh tn_acc K=%d [%-32s]: %.2f ms  %.1f TFLOP/s" % (K, name, ms, 2.0 * H * 4 * H * K / ms / 1e9))
    ops.tune_clear()

    if os.environ.get("MB_BF16"):
        ms = timeit(lambda: ops.lstm_forward(table, Wh, gates, h, c, T, N, H, 0, 4 * H, tok_gather=tok, flags=1))
        print("option LSTM fwd  bf16 operands: %.2f ms  (%.1f TFLOP/s-equivalent)" % (ms, fl / ms / 1e9))
        ms = timeit(lambda: ops.lstm_backward(Wh, gates, c, dcw, T, N, H, dh_last=dh_last, flags=1), iters=3, warm=1)
        print("option LSTM bwd  bf16 operands: %.2f ms  (%.1f TFLOP/s-equivalent)" % (ms, fl / ms / 1e9))
        ms = timeit(lambda: ops.gemm_tn_acc(hh, gg[N:], dWh, M=H, N=4 * H, K=K, flags=1), iters=3, warm=1)
        print("dWh tn_acc bf16 operands: %.2f ms  (%.1f TFLOP/s-equivalent)" % (ms, 2.0 * H * 4 * H * K / ms / 1e9))
    if only_big:
        return
    # encoder-sized recurrent steps (latency-bound)
    Ns, Ts = 200, 40
    toks = torch.randint(1, V, (Ts, Ns), device=dev, dtype=torch.int32, generator=g)
    xp = rnd(Ts, Ns, 4 * H)
    gs = torch.empty(Ts, Ns, 4 * H, device=dev)
    hs = torch.empty(Ts, Ns, H, device=dev)
    cs = torch.empty(Ts, Ns, H, device=dev)
    ms = timeit(lambda: ops.lstm_forward(xp, Wh, gs, hs, cs, Ts, Ns, H, Ns * 4 * H, 4 * H, tok_mask=toks), iters=10)
    print("encoder LSTM fwd T=40 N=200: %.3f ms  (%.1f us/step)" % (ms, ms * 1e3 / Ts))
    dcs = torch.empty(Ns, H, device=dev)
    dl = rnd(Ns, H)
    ms = timeit(lambda: ops.lstm_backward(Wh, gs, cs, dcs, Ts, Ns, H, dh_last=dl), iters=10)
    print("encoder LSTM bwd T=40 N=200: %.3f ms  (%.1f us/step)" % (ms, ms * 1e3 / Ts))

    # image-attention GEMM with fused loader/epilogue
    B, R, S2, Kc = 20, 10, 196, 512
    Nn = B * R
    pre = torch.tanh(rnd(B * S2, H))
    m1 = torch.randint(0, 2, (Nn * S2, H), device=dev, dtype=torch.uint8, generator=g)
    m2 = torch.randint(0, 2, (Nn * S2, Kc), device=dev, dtype=torch.uint8, generator=g)
    Wc, bc, qc = rnd(Kc, H) * 0.04, rnd(Kc), rnd(Nn, Kc)
    iqc = torch.empty(Nn * S2, Kc, device=dev)
    ms = timeit(lambda: ops.img_common_forward(pre, m1, Wc, bc, qc, m2, iqc, Nn, R, S2, H, Kc, 2.0))
    print("img_common fwd [39200x512x512]: %.3f ms  %.1f TFLOP/s" % (ms, 2.0 * Nn * S2 * H * Kc / ms / 1e9))

    wa, ba, u0 = rnd(Kc) * 0.05, rnd(1), rnd(Nn, H)
    patt, u1 = torch.empty(Nn, S2, device=dev), torch.empty(Nn, H, device=dev)
    ms = timeit(lambda: ops.img_att_forward(iqc, wa, ba, pre, m1, u0, patt, u1, Nn, R, S2, H, Kc, 2.0), iters=10)
    print("img_att fwd [200x196x512]: %.1f us" % (ms * 1e3))
    datt, dwa, dba = rnd(Nn, H), torch.zeros(Kc, device=dev), torch.zeros(1, device=dev)
    dqc, wk = torch.empty(Nn, Kc, device=dev), torch.empty(Nn, S2, device=dev)
    ms = timeit(lambda: ops.img_att_backward(iqc, wa, pre, m1, m2, patt, datt, dwa, dba, dqc, wk, Nn, R, S2, H, Kc, 2.0), iters=10)
    print("img_att bwd [200x196x512]: %.1f us" % (ms * 1e3))
    Qm, Hmm = rnd(Nn, H), rnd(Nn, H)
    mk = torch.triu(torch.ones(R, R, device=dev, dtype=torch.uint8), 1).repeat(B, 1, 1).contiguous()
    Pm, hatt = torch.empty(B, R, R, device=dev), torch.empty(Nn, H, device=dev)
    ms = timeit(lambda: ops.mn_attention_forward(Qm, Hmm, mk, Pm, hatt, B, R, H), iters=20)
    print("mn_att fwd [20x10x10x512]: %.1f us" % (ms * 1e3))
    dQ, dHm = torch.empty(Nn, H, device=dev), torch.empty(Nn, H, device=dev)
    ms = timeit(lambda: ops.mn_attention_backward(Qm, Hmm, Pm, datt, dQ, dHm, B, R, H), iters=20)
    print("mn_att bwd [20x10x10x512]: %.1f us" % (ms * 1e3))

    # plain GEMMs
    A = rnd(8000, E)
    Wx = rnd(E, 4 * H)
    bx = rnd(4 * H)
    out = torch.empty(8000, 4 * H, device=dev)
    ms = timeit(lambda: ops.gemm_nn(A, Wx, out, bias=bx))
    print("xproj nn [8000x2048x300]: %.3f ms  %.1f TFLOP/s" % (ms, 2.0 * 8000 * 2048 * 300 / ms / 1e9))
    A2, W2, o2 = rnd(200, H), rnd(H, H), torch.empty(200, H, device=dev)
    ms = timeit(lambda: ops.gemm_nt(A2, W2, o2, act=1), iters=20)
    print("linear nt [200x512x512]: %.1f us" % (ms * 1e3))

    # Adam
    n = 14166821
    w, gr, m, v = (torch.zeros(n, device=dev) for _ in range(4))
    ms = timeit(lambda: ops.clamp_adam(w, gr, m, v, 1e-3), iters=10)
    print("clamp+adam n=%d: %.3f ms  %.0f GB/s" % (n, ms, 28.0 * n / ms / 1e6))
    # option-table gradient: token sort + segmented row sum
    tokf = tok.view(-1)
    offset = torch.empty(V + 2, dtype=torch.int32, device=dev)
    work = torch.empty(2 * (V + 1), dtype=torch.int32, device=dev)
    perm = torch.empty(T * N, dtype=torch.int32, device=dev)
    ms = timeit(lambda: ops.token_sort(tokf, V + 1, offset, work, perm))
    print("token sort n=%d: %.3f ms" % (T * N, ms))
    dtab = torch.zeros(V + 1, 4 * H, device=dev)
    ms = timeit(lambda: ops.segment_rowsum_acc(gg, tokf, perm, dtab), iters=3, warm=1)
    print("segment rowsum [400000x2048]: %.3f ms  %.0f GB/s" % (ms, 4.0 * T * N * 4 * H / ms / 1e6))


if __name__ == "__main__":
    main()
