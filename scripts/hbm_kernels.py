"""achieved GB/s of every NON-GEMM kernel of the hot path at the headline shapes (BASELINE.json configs[3]:
20 dialogs x 10 rounds x 100 options, Tq = 20, Th = 40, 14x14x512, V = 11 322, H = 512), each run ALONE with HIP events
on its launch stream: algorithmic bytes (the bytes the operation has to move once: inputs read + outputs written, no
re-reads), average time, GB/s, fraction of the 8 TB/s HBM3E peak.  `python scripts/hbm_kernels.py [kernel_stats.txt]`:
with the per-kernel table of a rocprofv3 kernel trace of bench.py (scripts/rocpd_stats.py) the in-step average duration of the same kernels is
printed beside the stand-alone one.  Output is committed as profiles/r03_hbm_kernels.txt."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from visdial_amd import ops

PEAK = 8000.0  # GB/s, MI355X_MICROARCH.md


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3  # us


def in_step_table(path):
    """kernel name -> (calls, avg us) from the table scripts/rocpd_stats.py prints for a rocprofv3 kernel trace of bench.py
    (columns: kernel calls total_us avg_us min_us max_us %)"""
    out = {}
    if not path or not os.path.exists(path):
        return out
    for line in open(path):
        f = line.rstrip().rsplit(None, 6)
        if len(f) == 7:
            try:
                out[f[0].strip()] = (int(f[1]), float(f[3]))
            except ValueError:
                pass
    return out


def main():
    stats = in_step_table(sys.argv[1] if len(sys.argv) > 1 else None)
    dev = "cuda"
    B, R, O, Tq, Th, To, H, E, V, S2, K = 20, 10, 100, 20, 40, 20, 512, 300, 11322, 196, 512
    N, NO = B * R, B * R * O
    g = torch.Generator(device=dev).manual_seed(0)
    rnd = lambda *s: torch.randn(*s, device=dev, generator=g)
    ri = lambda lo, hi, *s: torch.randint(lo, hi, s, device=dev, dtype=torch.int32, generator=g)
    rows = []

    def add(name, kernel_substr, alg_bytes, fn, iters=20, note=''):
        us = timeit(fn, iters=iters)
        gbs = alg_bytes / us / 1e3
        subs = kernel_substr if isinstance(kernel_substr, (tuple, list)) else (kernel_substr,)
        ins = [(k, v) for k, v in stats.items() if any(x and x in k for x in subs)]
        ins_txt = ' | '.join('%s x%d avg %.1f us' % (k.split('(')[0][:40], v[0], v[1]) for k, v in ins) or '-'
        rows.append((name, alg_bytes / 1e6, us, gbs, gbs / PEAK, ins_txt, note))

    # --- LookupTableMaskZero forward / backward (question + history embeddings): [T*N x E] rows
    emb = rnd(V + 1, E)
    for tag, T in (('ques', Tq), ('hist', Th)):
        tok = ri(0, V + 1, T * N)
        out = torch.empty(T * N, E, device=dev)
        mask = torch.randint(0, 2, (T * N, E), device=dev, dtype=torch.uint8, generator=g)
        add('embed_gather %s [%d x %d] + dropout' % (tag, T * N, E), 'embed_gather', T * N * (E * 4 * 2 + E + 4),
            lambda: ops.embed_gather(emb, tok, out, mask=mask, scale=2.0))
        demb = torch.zeros(V + 1, E, device=dev)
        add('embed_scatter_acc %s [%d x %d] (float atomics)' % (tag, T * N, E), 'embed_scatter', T * N * (E * 4 * 2 + E + 4),
            lambda: ops.embed_scatter_acc(demb, tok, out, mask=mask, scale=2.0))
    # --- option-table gradient: counting sort of 400 000 tokens + segmented row sum of da [400 000 x 2048]
    tokf = ri(0, V + 1, To * NO)
    offset = torch.empty(V + 2, dtype=torch.int32, device=dev)
    work = torch.empty(2 * (V + 1), dtype=torch.int32, device=dev)
    perm = torch.empty(To * NO, dtype=torch.int32, device=dev)
    add('token_sort n=%d' % (To * NO), 'tok_', To * NO * 4 * 3, lambda: ops.token_sort(tokf, V + 1, offset, work, perm), iters=5)
    da = torch.empty(To * NO, 4 * H, device=dev).normal_(generator=g)
    dtab = torch.zeros(V + 1, 4 * H, device=dev)
    add('segment_rowsum [%d x %d] -> [%d x %d]' % (To * NO, 4 * H, V + 1, 4 * H), 'segment_rowsum',
        To * NO * 4 * H * 4 + (V + 1) * 4 * H * 4 * 2, lambda: ops.segment_rowsum_acc(da, tokf, perm, dtab), iters=3)
    del da, dtab
    # --- memory-network attention (MaskSoftMax over <= 10 facts)
    Q, Hm = rnd(N, H), rnd(N, H)
    mk = torch.triu(torch.ones(R, R, device=dev, dtype=torch.uint8), 1).repeat(B, 1, 1).contiguous()
    P, hatt = torch.empty(B, R, R, device=dev), torch.empty(N, H, device=dev)
    add('mn_att_fwd [20 x 10 x 10 x 512]', 'mn_att_fwd', N * H * 4 * 3 + N * R * 5, lambda: ops.mn_attention_forward(Q, Hm, mk, P, hatt, B, R, H),
        note='latency-bound')
    dQ, dHm, dh = torch.empty(N, H, device=dev), torch.empty(N, H, device=dev), rnd(N, H)
    add('mn_att_bwd', 'mn_att_bwd', N * H * 4 * 5 + N * R * 4, lambda: ops.mn_attention_backward(Q, Hm, P, dh, dQ, dHm, B, R, H),
        note='latency-bound')
    # --- image attention (SoftMax over 196 regions + weighted sum), forward and backward
    pre = torch.tanh(rnd(B * S2, H))
    m1 = torch.randint(0, 2, (N * S2, H), device=dev, dtype=torch.uint8, generator=g)
    m2 = torch.randint(0, 2, (N * S2, K), device=dev, dtype=torch.uint8, generator=g)
    iqc = rnd(N * S2, K)
    wa, ba, u0 = rnd(K) * 0.05, rnd(1), rnd(N, H)
    patt, u1 = torch.empty(N, S2, device=dev), torch.empty(N, H, device=dev)
    fwd_bytes = N * S2 * K * 4 + B * S2 * H * 4 + N * S2 * H + N * H * 8 + N * S2 * 4
    add('img_att forward (score + softmax + weighted sum)', ('img_att_score', 'img_att_wsum'), fwd_bytes,
        lambda: ops.img_att_forward(iqc, wa, ba, pre, m1, u0, patt, u1, N, R, S2, H, K, 2.0))
    datt, dwa, dba = rnd(N, H), torch.zeros(K, device=dev), torch.zeros(1, device=dev)
    dqc, wk = torch.empty(N, K, device=dev), torch.empty(N, S2, device=dev)
    bwd_bytes = N * S2 * K * 4 * 2 + B * S2 * H * 4 + N * S2 * (H + K) + N * H * 4 + N * K * 4 + N * S2 * 8
    add('img_att backward (dscore + dz in place)', ('img_att_dscore', 'img_att_dz'), bwd_bytes,
        lambda: ops.img_att_backward(iqc, wa, pre, m1, m2, patt, datt, dwa, dba, dqc, wk, N, R, S2, H, K, 2.0))
    # --- option scoring + CrossEntropy (forward + backward in one kernel)
    optH, enc = rnd(NO, H), rnd(N, H)
    gt = ri(0, O, N)
    scores, lr = torch.empty(N, O, device=dev), torch.empty(N, device=dev)
    dOpt, dEnc = torch.empty(NO, H, device=dev), torch.empty(N, H, device=dev)
    add('score_ce [200 x 100 x 512] fwd + bwd', 'score_ce', NO * H * 4 * 2 + N * H * 8 + N * O * 4,
        lambda: ops.score_ce(optH, enc, scores, N, O, H, gt=gt, loss_rows=lr, dOptH=dOpt, dEnc=dEnc, gscale=1.0 / N))
    rk = torch.empty(N, O, device=dev, dtype=torch.int32)
    add('ranks [200 x 100]', 'ranks_kernel', N * O * 8, lambda: ops.ranks(scores, rk, N, O), note='latency-bound')
    # --- clamp + Adam over the flat parameter vector
    n = 14166821
    w, gr, m, v = (torch.zeros(n, device=dev) for _ in range(4))
    add('clamp_adam n=%d' % n, 'clamp_adam', 28 * n, lambda: ops.clamp_adam(w, gr, m, v, 1e-3))
    # --- dropout mask generation (device counter-based generator): the two [N x 196 x 512] byte masks
    mk2 = torch.empty(N * S2 * H, device=dev, dtype=torch.uint8)
    add('dropout_mask [%d bytes]' % mk2.numel(), 'dropout_mask', mk2.numel(), lambda: ops.dropout_mask(mk2, 1234, 0.5))
    # --- generative decoder head: LogSoftMax + NLL over [Ta*N x V] (configs[1]; not in the headline step)
    Ta = 21
    logits = rnd(Ta * N, V)
    tin, tgt = ri(1, V, Ta * N), ri(1, V, Ta * N)
    lrow = torch.empty(Ta * N, device=dev)
    add('logsoftmax_nll [%d x %d] (gen decoder) fwd + grad in place' % (Ta * N, V), 'logsoftmax_nll', Ta * N * V * 4 * 2,
        lambda: ops.logsoftmax_nll(logits, V, tin, tgt, lrow, write_grad=True), iters=5)

    print('%-62s %10s %10s %9s %7s   %s' % ('kernel (alone, HIP events)', 'alg MB', 'avg us', 'GB/s', 'of 8TB/s', 'in-step (rocprofv3 kernel stats)'))
    for name, mb, us, gbs, frac, ins, note in rows:
        print('%-62s %10.2f %10.1f %9.0f %7.3f   %s%s' % (name, mb, us, gbs, frac, ins, ('   [' + note + ']') if note else ''))


if __name__ == "__main__":
    main()
