"""Operator-level parity: every C-ABI entry point (called through ctypes on device buffers) against
the numpy oracle on seeded inputs.  fp32 tolerance: rel-L2 <= 1e-5 for single ops (north_star: 1e-4
end-to-end); integer outputs bit-exact."""
import numpy as np
import pytest
import torch

from oracle import visdial_oracle as vo

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from visdial_amd import ops as o
    return o


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def relerr(got, ref):
    got = got.detach().cpu().numpy().astype(np.float64) if torch.is_tensor(got) else np.asarray(got, np.float64)
    ref = np.asarray(ref, np.float64)
    return float(np.linalg.norm(got - ref) / max(np.linalg.norm(ref), 1e-30))


def f32(rng, *shape):
    return rng.randn(*shape).astype(np.float32)


# asymmetric operands everywhere (guide: symmetric inputs hide transposes)
# (M >= 1024 with K % 16 == 0 runs on the LDS-DMA pipeline: 4 / 5 / 128 K tiles, partial row and column tiles)
@pytest.mark.parametrize("M,N,K", [(200, 512, 512), (3920, 512, 512), (260, 300, 300), (33, 40, 4), (1, 512, 2048),
                                   (1500, 300, 64), (1025, 44, 80), (11323, 300, 2048),
                                   (5000, 2048, 300)])
@pytest.mark.parametrize("act", [0, 1])
def test_gemm_nt(ops, M, N, K, act):
    rng = np.random.RandomState(M + N + K)
    A, W, b = f32(rng, M, K), f32(rng, N, K) * 0.1, f32(rng, N)
    C = torch.zeros(M, N, device="cuda")
    ops.gemm_nt(dev(A), dev(W), C, bias=dev(b), act=act)
    ref = A.astype(np.float64) @ W.astype(np.float64).T + b
    if act:
        ref = np.tanh(ref)
    assert relerr(C, ref) < 1e-5
    C0 = f32(rng, M, N)
    Cd = dev(C0)
    ops.gemm_nt(dev(A), dev(W), Cd, bias=None, act=0, accumulate=True)
    assert relerr(Cd, C0 + A.astype(np.float64) @ W.astype(np.float64).T) < 1e-5


@pytest.mark.parametrize("M,N,K", [(11323, 300, 2048), (1500, 512, 304), (1100, 130, 4096)])
def test_gemm_nt_atomic_accumulate_splits_k(ops, M, N, K):
    """accumulate = 2 (hardware float atomics; dEmb += dTable * Wx^T beside the encoder's scatters), ragged last K slice included"""
    rng = np.random.RandomState(M + N + K + 7)
    A, W = f32(rng, M, K), f32(rng, N, K) * 0.1
    C0 = f32(rng, M, N)
    Cd = dev(C0)
    ref = C0 + A.astype(np.float64) @ W.astype(np.float64).T
    ops.gemm_nt(dev(A), dev(W), Cd, bias=None, act=0, accumulate=2)
    assert relerr(Cd, ref) < 1e-5
    Ce = dev(C0)
    ops.gemm_nt(dev(A), dev(W), Ce, bias=None, act=0, accumulate=2)
    assert relerr(Ce, ref) < 1e-5 and relerr(Cd, Ce.cpu().numpy()) < 1e-5


@pytest.mark.parametrize("M,N,K", [(8000, 2048, 300), (200, 2048, 512), (130, 300, 2048), (51, 2048, 12), (4000, 512, 512)])
def test_gemm_nn(ops, M, N, K):
    rng = np.random.RandomState(M + N + K + 1)
    A, B, b = f32(rng, M, K), f32(rng, K, N) * 0.1, f32(rng, N)
    C = torch.zeros(M, N, device="cuda")
    ops.gemm_nn(dev(A), dev(B), C, bias=dev(b))
    assert relerr(C, A.astype(np.float64) @ B.astype(np.float64) + b) < 1e-5


@pytest.mark.parametrize("M,N,K", [(300, 2048, 8000), (512, 512, 200), (300, 2048, 11323), (512, 2048, 40000), (12, 128, 37)])
def test_gemm_tn_acc(ops, M, N, K):
    rng = np.random.RandomState(M + N + K + 2)
    A, B = f32(rng, K, M), f32(rng, K, N) * 0.1
    C0 = f32(rng, M, N)
    C = dev(C0)
    ops.gemm_tn_acc(dev(A), dev(B), C)
    assert relerr(C, C0 + A.astype(np.float64).T @ B.astype(np.float64)) < 1e-5


@pytest.mark.parametrize("M,N,R,K", [(512, 2048, 8000, 4431), (300, 2048, 8000, 4431), (36, 384, 135, 77), (96, 384, 60, 1),
                                     (512, 2048, 4000, 4000)])
def test_gemm_tn_rows_acc(ops, M, N, R, K):
    """the weight-gradient contraction over an explicit list of (row of A, row of B) pairs (non-pad (t, row) pairs of
    a maskZero recurrence): ordered and repeated indices, different lists for the two operands, ragged K"""
    import torch
    rng = np.random.RandomState(M + N + K)
    A, B = f32(rng, R, M), f32(rng, R, N) * 0.1
    ra = np.sort(rng.choice(R, size=K, replace=K > R)).astype(np.int32)
    rb = np.maximum(ra - rng.randint(0, 3), 0).astype(np.int32)
    C0 = f32(rng, M, N)
    C = dev(C0)
    ops.gemm_tn_rows_acc(dev(A), torch.from_numpy(ra).cuda(), dev(B), torch.from_numpy(rb).cuda(), C)
    ref = C0 + A[ra].astype(np.float64).T @ B[rb].astype(np.float64)
    assert relerr(C, ref) < 1e-5


@pytest.mark.parametrize("M,N,ld", [(8001, 2048, None), (9800, 512, None), (11323, 2048, None), (63, 2048, None), (4001, 300, 304),
                                    (777, 11322, 11324), (65, 4, None), (1, 2048, None)])
def test_colsum(ops, M, N, ld):
    """bias gradients: the float4 kernel (N % 4 == 0, aligned rows, M >= 64: ragged row counts, a narrowed view with ld > N) and the
    scalar one (N % 4 != 0, short M)"""
    rng = np.random.RandomState(0)
    X = f32(rng, M, ld or N)
    out0 = f32(rng, N)
    out = dev(out0)
    ops.colsum_acc(dev(X), out, M=M, N=N, ld=ld or N)
    assert relerr(out, out0 + X[:, :N].astype(np.float64).sum(0)) < 1e-5


# N >= 2048 rows take the LDS-DMA pipeline: H = 32 / 64 / 96 give 2, 4 and 6 K tiles in the forward step (fewer
# than, equal to and more than the 3 LDS buffers), ragged last row tile, masked and gathered input rows
# (6, 2300, 512, table): the headline's H = 512 table-gather mode.
@pytest.mark.parametrize("T,N,H,masked,table", [(5, 200, 64, True, False), (4, 2500, 64, False, True),
                                                (3, 2100, 512, True, False), (6, 37, 32, True, False),
                                                (3, 2049, 32, False, True), (3, 2177, 96, True, True),
                                                (6, 2300, 512, False, True)])
def test_lstm_forward_backward(ops, T, N, H, masked, table):
    _lstm_forward_backward(ops, T, N, H, masked, table)


def _lstm_forward_backward(ops, T, N, H, masked, table, flags=0, errors=None):
    rng = np.random.RandomState(T * 1000 + N + H)
    D = 20
    V = 30
    emb = f32(rng, V + 1, D)
    emb[0] = 0
    tok = rng.randint(0, V + 1, size=(T, N)).astype(np.int32)
    if masked:   # right-aligned: leading pads
        lens = rng.randint(0, T + 1, size=N)
        for n in range(N):
            tok[:T - lens[n], n] = 0
            tok[T - lens[n]:, n] = np.maximum(tok[T - lens[n]:, n], 1)
    x = emb[tok]
    W = (f32(rng, D + H, 4 * H) / np.sqrt(D + H)).astype(np.float32)
    b = f32(rng, 4 * H) * 0.1
    h_ref, c_ref, g_ref = vo.lstm_forward(x.astype(np.float64), W.astype(np.float64), b.astype(np.float64),
                                          tok if masked else None)
    Wh = dev(W[D:])
    gates = torch.empty(T, N, 4 * H, device="cuda")
    h = torch.empty(T, N, H, device="cuda")
    c = torch.empty(T, N, H, device="cuda")
    tok_d = dev(tok)
    if table:
        tab = (emb.astype(np.float64) @ W[:D].astype(np.float64) + b).astype(np.float32)
        ops.lstm_forward(dev(tab), Wh, gates, h, c, T, N, H, 0, 4 * H, tok_gather=tok_d,
                         tok_mask=tok_d if masked else None, flags=flags)
    else:
        xp = (x.reshape(T * N, D).astype(np.float64) @ W[:D].astype(np.float64) + b).astype(np.float32)
        ops.lstm_forward(dev(xp), Wh, gates, h, c, T, N, H, N * 4 * H, 4 * H, tok_mask=tok_d if masked else None, flags=flags)
    torch.cuda.synchronize()
    if errors is not None:
        errors.update(h=relerr(h, h_ref), c=relerr(c, c_ref), gates=relerr(gates, g_ref))
    else:
        assert relerr(h, h_ref) < 1e-5 and relerr(c, c_ref) < 1e-5 and relerr(gates, g_ref) < 1e-5

    dh_seq = f32(rng, T, N, H)
    dh_last = f32(rng, N, H)
    _, dW_ref, _, dh0_ref, dc0_ref, da_ref = vo.lstm_backward(
        x.astype(np.float64), W.astype(np.float64), g_ref, h_ref, c_ref, dh_seq=dh_seq.astype(np.float64),
        dh_last=dh_last.astype(np.float64), return_da=True)
    # feed the device its own forward state (already checked above)
    dc_work = torch.empty(N, H, device="cuda")
    dh0 = torch.empty(N, H, device="cuda")
    dWh0 = f32(rng, H, 4 * H)
    dWh = dev(dWh0)            # accumulated INTO (accGradParameters): recurrence + trailing weight gradient
    if errors is not None:      # the error table measures the BACKWARD arithmetic alone: feed it the exact forward state
        gates.copy_(dev(g_ref.astype(np.float32)))
        c.copy_(dev(c_ref.astype(np.float32)))
        h.copy_(dev(h_ref.astype(np.float32)))
    ops.lstm_backward(Wh, gates, c, dc_work, T, N, H, dh_seq=dev(dh_seq), dh_last=dev(dh_last), dh0=dh0,
                      h_seq=h, dWh=dWh, flags=flags)
    torch.cuda.synchronize()
    if errors is not None:
        errors.update(da=relerr(gates, da_ref), dc0=relerr(dc_work, dc0_ref), dh0=relerr(dh0, dh0_ref), dWh=relerr(dWh, dWh0 + dW_ref[D:]))
        return
    assert relerr(gates, da_ref) < 2e-5
    assert relerr(dc_work, dc0_ref) < 2e-5
    assert relerr(dh0, dh0_ref) < 2e-5
    assert relerr(dWh, dWh0 + dW_ref[D:]) < 2e-5


@pytest.mark.parametrize("T,N,H,masked,table", [(4, 2500, 64, False, True), (3, 2100, 512, True, False), (3, 2049, 32, False, True),
                                                (3, 2177, 96, True, True), (6, 2300, 512, False, True)])
def test_lstm_forward_backward_split9(ops, T, N, H, masked, table):
    """`lstmPrecision = split9` (csrc/split_core.h): the recurrent products as the EXACT three-way bf16 split of both operands,
    nine bf16 MFMAs per fp32 one -- held to the SAME 1e-5 / 2e-5 bounds against fp64 as the fp32 MFMA path above"""
    _lstm_forward_backward(ops, T, N, H, masked, table, flags=ops.FLAG_SPLIT9)


def test_split_error_table(ops):
    """rel-L2 error against fp64 of the recurrence outputs at the headline's H = 512, per arithmetic: exact fp32 MFMA, split9 (all
    nine products), split6 (i + j <= 2), split3 (i + j <= 1), bf16.  split9 must sit with fp32; split6 / split3 / bf16 are printed as
    data (only split9 may stand in for fp32).  The table goes to gpurun_out/r05_split_errors.txt."""
    import os
    rows = []
    for name, flags in (('fp32 (v_mfma_f32_32x32x2_f32)', 0), ('split9', ops.FLAG_SPLIT9), ('split6', ops.FLAG_SPLIT6),
                        ('split3', ops.FLAG_SPLIT3), ('bf16', ops.FLAG_BF16)):
        e = {}
        _lstm_forward_backward(ops, 6, 2300, 512, False, True, flags=flags, errors=e)
        rows.append((name, e))
    keys = ('h', 'c', 'gates', 'da', 'dc0', 'dh0', 'dWh')
    lines = ['# rel-L2 error vs the fp64 oracle, T = 6, N = 2300, H = 512 (table-gather mode); forward from the same inputs, backward from the EXACT forward state',
             '%-32s ' % 'arithmetic' + ' '.join('%10s' % k for k in keys)]
    for name, e in rows:
        lines.append('%-32s ' % name + ' '.join('%10.2e' % e[k] for k in keys))
    text = '\n'.join(lines)
    print(text)
    os.makedirs('gpurun_out', exist_ok=True)
    open('gpurun_out/r05_split_errors.txt', 'w').write(text + '\n')
    err = {name: e for name, e in rows}
    f32, s9, s6, s3, b16 = (err[n] for n in ('fp32 (v_mfma_f32_32x32x2_f32)', 'split9', 'split6', 'split3', 'bf16'))
    for k in keys:
        assert s9[k] < 2e-5 and s9[k] < 3 * f32[k] + 1e-7, (k, s9[k], f32[k])          # fp32-grade
    assert s3['h'] > 3 * s9['h'] and b16['h'] > 10 * s3['h']                            # and the ladder is real


def test_bf16_weight_gradient_on_producer_shadows(ops):
    """BASELINE.json configs[4] (opt-in bf16 option recurrence): the LSTM step kernels of a bf16 pass also write bf16
    copies of h and da, and the dWh contraction multiplies those copies directly (LDS-DMA + ds_read_b64_tr_b16, no
    conversion while staging).  Checked against the fp32 product of the SAME h / da the pass produced: the only difference
    is the bf16 rounding of the operands (rel-L2 < 1e-2).  K = 5 x 4168 rows is not a multiple of 32: the tail rows take
    the staging kernel."""
    T, N, H, V = 6, 4168, 128, 40
    rng = np.random.RandomState(5)
    Wh = dev((f32(rng, H, 4 * H) / np.sqrt(H)).astype(np.float32))
    tab = dev(f32(rng, V + 1, 4 * H) * 0.5)
    tok = dev(rng.randint(0, V + 1, size=(T, N)).astype(np.int32))
    dh_last = dev(f32(rng, N, H))
    gates = torch.empty(T, N, 4 * H, device="cuda")
    h = torch.empty(T, N, H, device="cuda")
    c = torch.empty(T, N, H, device="cuda")
    dc = torch.empty(N, H, device="cuda")
    ops.lstm_forward(tab, Wh, gates, h, c, T, N, H, 0, 4 * H, tok_gather=tok, flags=1)
    ops.lstm_backward(Wh, gates, c, dc, T, N, H, dh_last=dh_last, flags=1)
    dWh = torch.zeros(H, 4 * H, device="cuda")
    K = (T - 1) * N
    ops.gemm_tn_acc(h.view(T * N, H), gates.view(T * N, 4 * H)[N:], dWh, M=H, N=4 * H, K=K, flags=1)
    torch.cuda.synchronize()
    ref = h.view(T * N, H)[:K].double().T @ gates.view(T * N, 4 * H)[N:].double()
    assert float((dWh.double() - ref).norm() / ref.norm()) < 1e-2


def test_bf16_shadow_registry_drops_overlapping_entries(ops):
    """ADVICE r5 (csrc/api.hip): a bf16 pass registers shadows for its h / gates tensors by BASE ADDRESS.  When a second, smaller recurrence
    lives INSIDE the range of an earlier one (the allocator handing out a sub-block of a freed tensor), the earlier entries must stop matching:
    the dWh contraction of the second pass has to multiply the second pass's bf16 copies, not the first pass's bytes at the same offset."""
    H, V = 128, 40
    rng = np.random.RandomState(9)
    Wh = dev((f32(rng, H, 4 * H) / np.sqrt(H)).astype(np.float32))
    tab = dev(f32(rng, V + 1, 4 * H) * 0.5)

    def run(T, N, hbuf, gbuf, off_rows, seed):
        r = np.random.RandomState(seed)
        h = hbuf[off_rows * H: off_rows * H + T * N * H].view(T, N, H)
        gates = gbuf[off_rows * 4 * H: off_rows * 4 * H + T * N * 4 * H].view(T, N, 4 * H)
        c = torch.empty(T, N, H, device="cuda")
        dc = torch.empty(N, H, device="cuda")
        tok = dev(r.randint(0, V + 1, size=(T, N)).astype(np.int32))
        dh_last = dev(f32(r, N, H))
        ops.lstm_forward(tab, Wh, gates, h, c, T, N, H, 0, 4 * H, tok_gather=tok, flags=1)
        ops.lstm_backward(Wh, gates, c, dc, T, N, H, dh_last=dh_last, flags=1)
        return h, gates

    T1, N1 = 3, 4096
    hbuf = torch.empty(T1 * N1 * H, device="cuda")
    gbuf = torch.empty(T1 * N1 * 4 * H, device="cuda")
    run(T1, N1, hbuf, gbuf, 0, 1)                                # registers [X, X + T1 N1 H) and the gates range
    T2, N2, off = 3, 2048, 1024                                  # a smaller pass at X + offset, inside both old ranges
    h, gates = run(T2, N2, hbuf, gbuf, off, 2)
    dWh = torch.zeros(H, 4 * H, device="cuda")
    K = (T2 - 1) * N2
    ops.gemm_tn_acc(h.view(T2 * N2, H), gates.view(T2 * N2, 4 * H)[N2:], dWh, M=H, N=4 * H, K=K, flags=1)
    torch.cuda.synchronize()
    ref = h.view(T2 * N2, H)[:K].double().T @ gates.view(T2 * N2, 4 * H)[N2:].double()
    assert float((dWh.double() - ref).norm() / ref.norm()) < 1e-2


def test_embed_gather_scatter(ops):
    rng = np.random.RandomState(1)
    V, E, rows = 40, 300, 1234
    emb = f32(rng, V + 1, E)
    emb[0] = 0
    tok = rng.randint(0, V + 1, size=rows).astype(np.int32)
    mask = (rng.rand(rows, E) > 0.5).astype(np.uint8)
    out = torch.empty(rows, E, device="cuda")
    ops.embed_gather(dev(emb), dev(tok), out, mask=dev(mask), scale=2.0)
    np.testing.assert_array_equal(out.cpu().numpy(), emb[tok] * mask * 2.0)
    ops.embed_gather(dev(emb), dev(tok), out)
    np.testing.assert_array_equal(out.cpu().numpy(), emb[tok])
    dx = f32(rng, rows, E)
    demb = torch.zeros(V + 1, E, device="cuda")
    ops.embed_scatter_acc(demb, dev(tok), dev(dx), mask=dev(mask), scale=2.0)
    ref = np.zeros((V + 1, E))
    vo.lookup_backward(ref, tok, dx.astype(np.float64) * mask * 2.0)
    assert relerr(demb, ref) < 1e-5


def test_token_sort_and_segment_sum(ops):
    rng = np.random.RandomState(2)
    V1, n, ncol = 57, 5000, 256
    tok = rng.randint(0, V1, size=n).astype(np.int32)
    tok[rng.rand(n) < 0.5] = 0     # a dominant pad token
    offset = torch.empty(V1 + 1, dtype=torch.int32, device="cuda")
    work = torch.empty(2 * V1, dtype=torch.int32, device="cuda")
    perm = torch.empty(n, dtype=torch.int32, device="cuda")
    tok_d = dev(tok)
    ops.token_sort(tok_d, V1, offset, work, perm)
    pm = perm.cpu().numpy()
    assert sorted(pm.tolist()) == list(range(n))                       # a permutation
    assert np.all(np.diff(tok[pm]) >= 0)                               # sortedness
    np.testing.assert_array_equal(offset.cpu().numpy(), np.concatenate([[0], np.cumsum(np.bincount(tok, minlength=V1))]))
    X = f32(rng, n, ncol)
    out = torch.zeros(V1, ncol, device="cuda")
    ops.segment_rowsum_acc(dev(X), tok_d, perm, out)
    ref = np.zeros((V1, ncol))
    np.add.at(ref, tok, X.astype(np.float64))
    assert relerr(out, ref) < 1e-5


def test_dropout_and_glue(ops):
    n = 1000003
    mask = torch.empty(n + 1, dtype=torch.uint8, device="cuda")[:n]
    ops.dropout_mask(mask, 1234, 0.5)
    m = mask.cpu().numpy()
    assert set(np.unique(m).tolist()) <= {0, 1} and abs(m.mean() - 0.5) < 5e-3
    m2 = torch.empty_like(mask)
    ops.dropout_mask(m2, 1234, 0.5)
    assert torch.equal(mask, m2)                                        # replayable
    ops.dropout_mask(m2, 1235, 0.5)
    assert not torch.equal(mask, m2)
    rng = np.random.RandomState(3)
    x, y = f32(rng, 4097), f32(rng, 4097)
    mk = (rng.rand(4097) > 0.5).astype(np.uint8)
    out = torch.empty(4097, device="cuda")
    ops.dropout_apply(dev(x), dev(mk), out, 2.0)
    np.testing.assert_array_equal(out.cpu().numpy(), x * mk * 2.0)
    t = np.tanh(x)
    ops.tanh_backward(dev(y), dev(t), out)
    np.testing.assert_allclose(out.cpu().numpy(), y * (1 - t * t), rtol=1e-6, atol=1e-7)
    ops.axpby(dev(x), dev(y), out, 0.5, -2.0)
    np.testing.assert_allclose(out.cpu().numpy(), 0.5 * x - 2.0 * y, rtol=1e-6)


def test_mn_attention(ops):
    rng = np.random.RandomState(4)
    B, R, H = 7, 10, 512
    q, h = f32(rng, B, R, H) * 0.2, f32(rng, B, R, H) * 0.2
    mask = vo.causal_mask(B, R)
    p_ref, hatt_ref = vo.mn_attention_forward(q.astype(np.float64), h.astype(np.float64), mask)
    P = torch.empty(B, R, R, device="cuda")
    hatt = torch.empty(B, R, H, device="cuda")
    ops.mn_attention_forward(dev(q), dev(h), dev(mask), P, hatt, B, R, H)
    assert relerr(P, p_ref) < 1e-5 and relerr(hatt, hatt_ref) < 1e-5
    assert float(P.cpu().numpy()[:, 0, 1:].max()) == 0.0               # hidden facts get exactly zero
    dh = f32(rng, B, R, H)
    dq_ref, dh_ref = vo.mn_attention_backward(q.astype(np.float64), h.astype(np.float64), p_ref, dh.astype(np.float64))
    dQ = torch.empty(B, R, H, device="cuda")
    dH = torch.empty(B, R, H, device="cuda")
    ops.mn_attention_backward(dev(q), dev(h), P, dev(dh), dQ, dH, B, R, H)
    assert relerr(dQ, dq_ref) < 1e-5 and relerr(dH, dh_ref) < 1e-5


@pytest.mark.parametrize("use_drop", [False, True])
def test_image_attention(ops, use_drop):
    rng = np.random.RandomState(5)
    B, R, S2, H, Kc = 3, 4, 49, 64, 96
    N = B * R
    pre = np.tanh(f32(rng, B * S2, H))
    m1 = (rng.rand(N * S2, H) > 0.5).astype(np.uint8) if use_drop else None
    m2 = (rng.rand(N * S2, Kc) > 0.5).astype(np.uint8) if use_drop else None
    Wc, bc = f32(rng, Kc, H) * 0.1, f32(rng, Kc) * 0.1
    qc = f32(rng, N, Kc) * 0.5
    wa, ba = f32(rng, Kc) * 0.3, f32(rng, 1)
    u0 = f32(rng, N, H)
    sc = 2.0 if use_drop else 1.0
    d64 = np.float64
    img_tr = np.repeat(pre.reshape(B, 1, S2, H), R, 1).reshape(N * S2, H).astype(d64)
    if use_drop:
        img_tr = img_tr * m1 * 2.0
    t_iqc = np.tanh(img_tr @ Wc.astype(d64).T + bc + np.repeat(qc.astype(d64), S2, 0))
    iqc_ref = t_iqc * m2 * 2.0 if use_drop else t_iqc
    score = (iqc_ref @ wa.astype(d64) + ba[0]).reshape(N, S2)
    e = np.exp(score - score.max(1, keepdims=True))
    p_ref = e / e.sum(1, keepdims=True)
    att = np.einsum('ns,nsh->nh', p_ref, img_tr.reshape(N, S2, H))
    u1_ref = att + u0

    pre_d, m1_d, m2_d = dev(pre), (dev(m1) if use_drop else None), (dev(m2) if use_drop else None)
    iqc = torch.empty(N * S2, Kc, device="cuda")
    ops.img_common_forward(pre_d, m1_d, dev(Wc), dev(bc), dev(qc), m2_d, iqc, N, R, S2, H, Kc, sc)
    assert relerr(iqc, iqc_ref) < 1e-5
    p = torch.empty(N, S2, device="cuda")
    u1 = torch.empty(N, H, device="cuda")
    ops.img_att_forward(iqc, dev(wa), dev(ba), pre_d, m1_d, dev(u0), p, u1, N, R, S2, H, Kc, sc)
    assert relerr(p, p_ref) < 1e-5 and relerr(u1, u1_ref) < 1e-5

    datt = f32(rng, N, H)
    dp = np.einsum('nh,nsh->ns', datt.astype(d64), img_tr.reshape(N, S2, H))
    dscore = p_ref * (dp - (p_ref * dp).sum(1, keepdims=True))
    dwa_ref = np.einsum('ns,nsk->k', dscore, iqc_ref.reshape(N, S2, Kc))
    diqc = dscore[:, :, None] * wa.astype(d64)[None, None, :]
    dz_ref = (diqc * (m2.reshape(N, S2, Kc) * 2.0 if use_drop else 1.0)) * (1 - t_iqc.reshape(N, S2, Kc) ** 2)
    dqc_ref = dz_ref.sum(1)
    dwa = torch.zeros(Kc, device="cuda")
    dba = torch.zeros(1, device="cuda")
    dqc = torch.empty(N, Kc, device="cuda")
    work = torch.empty(N, S2, device="cuda")
    ops.img_att_backward(iqc, dev(wa), pre_d, m1_d, m2_d, p, dev(datt), dwa, dba, dqc, work, N, R, S2, H, Kc, sc)
    assert relerr(iqc, dz_ref.reshape(N * S2, Kc)) < 1e-5
    assert relerr(dwa, dwa_ref) < 1e-5 and relerr(dqc, dqc_ref) < 1e-5
    assert abs(float(dba.item()) - dscore.sum()) < 1e-4
    dimg = p_ref[:, :, None] * datt.astype(d64)[:, None, :] + (dz_ref.reshape(N * S2, Kc) @ Wc.astype(d64)).reshape(N, S2, H)
    if use_drop:
        dimg = dimg * m1.reshape(N, S2, H) * 2.0
    dpre_ref = dimg.reshape(B, R, S2, H).sum(1).reshape(B * S2, H)
    dpre = torch.zeros(B * S2, H, device="cuda")
    ops.img_tr_backward(iqc, dev(Wc), p, dev(datt), m1_d, dpre, N, R, S2, H, Kc, sc)
    assert relerr(dpre, dpre_ref) < 1e-5
    dWc = torch.zeros(Kc, H, device="cuda")
    ops.img_common_wgrad(iqc, pre_d, m1_d, dWc, N, R, S2, H, Kc, sc)
    assert relerr(dWc, dz_ref.reshape(N * S2, Kc).T @ img_tr) < 1e-5


def test_score_ce_and_ranks(ops):
    rng = np.random.RandomState(6)
    N, O, H = 23, 100, 512
    optH, enc = f32(rng, N, O, H) * 0.2, f32(rng, N, H) * 0.2
    gt = rng.randint(0, O, size=N).astype(np.int32)
    s_ref = np.einsum('noh,nh->no', optH.astype(np.float64), enc.astype(np.float64))
    loss_ref, ds_ref, rows_ref = vo.cross_entropy(s_ref, gt)
    scores = torch.empty(N, O, device="cuda")
    rows = torch.empty(N, device="cuda")
    dO = torch.empty(N, O, H, device="cuda")
    dE = torch.empty(N, H, device="cuda")
    ops.score_ce(dev(optH), dev(enc), scores, N, O, H, gt=dev(gt), loss_rows=rows, dOptH=dO, dEnc=dE, gscale=1.0 / N)
    assert relerr(scores, s_ref) < 1e-5 and relerr(rows, rows_ref) < 1e-5
    assert abs(float(rows.mean().item()) - loss_ref) < 1e-5
    assert relerr(dO, ds_ref[:, :, None] * enc.astype(np.float64)[:, None, :]) < 1e-5
    assert relerr(dE, np.einsum('no,noh->nh', ds_ref, optH.astype(np.float64))) < 1e-5
    # ranks: bit-exact against the oracle on the DEVICE scores (incl. forced ties)
    sc = scores.clone()
    sc[0, 5] = sc[0, 9]
    sc[1, :] = 0.25
    rk = torch.empty(N, O, dtype=torch.int32, device="cuda")
    ops.ranks(sc, rk, N, O)
    np.testing.assert_array_equal(rk.cpu().numpy(), vo.compute_ranks(sc.cpu().numpy()))


def test_clamp_adam(ops):
    rng = np.random.RandomState(7)
    n = 100003
    w, g = f32(rng, n), f32(rng, n) * 4
    wd, gd = dev(w), dev(g)
    m = torch.zeros(n, device="cuda")
    v = torch.zeros(n, device="cuda")
    st = {}
    wr = w.astype(np.float64)
    lr = 1e-3
    for t in range(1, 4):
        step = lr * np.sqrt(1 - 0.999 ** t) / (1 - 0.9 ** t)
        ops.clamp_adam(wd, gd, m, v, step)
        wr, gc = vo.clamp_adam(wr, g.astype(np.float64), st, lr)
        gd.copy_(dev(g))
    assert relerr(wd, wr) < 1e-6
    assert relerr(m, st["m"]) < 5e-5 and relerr(v, st["v"]) < 5e-5


def test_error_reporting(ops):
    from visdial_amd import _lib
    with pytest.raises(_lib.VisdialHipError):
        A = torch.zeros(4, 6, device="cuda")
        ops.gemm_nt(A, A, torch.zeros(4, 4, device="cuda"), K=6)       # K % 4 != 0 -> rejected loudly


@pytest.mark.parametrize("arith", ['fp32', 'bf16'])
def test_lstm2_wavefront_matches_oracle(ops, arith):
    """two stacks (T=7 and T=4) x two layers through the grouped skewed-wavefront drivers.  fp32: the C-ABI entry points (fp32 MFMA),
    1e-5 / 2e-5 against the fp64 oracle.  bf16: the ticks of a bf16 pass of the model-level runtime (configs[4]; csrc/lstm.hip
    `vd_lstm2_forward_p(flags = VD_FLAG_BF16)`: both operands of every recurrent product rounded to bf16 while staged, fp32 accumulate,
    fp32 state) within the bf16 bound of the config (1e-2 states, 2e-2 gate gradients) and measurably different from fp32."""
    if arith == 'bf16':
        fwd = lambda st, H: ops.lstm2_pass(st, H, ops.FLAG_BF16)
        bwd = lambda st, H: ops.lstm2_pass(st, H, ops.FLAG_BF16, backward=True)
        tol_f, tol_b = 1e-2, 2e-2
    else:
        fwd, bwd, tol_f, tol_b = ops.lstm2_forward, ops.lstm2_backward, 1e-5, 2e-5
    rng = np.random.RandomState(11)
    H, D, V = 64, 20, 30
    stacks = []
    for T, N in ((7, 70), (4, 70)):
        emb = f32(rng, V + 1, D); emb[0] = 0
        tok = rng.randint(1, V + 1, size=(T, N)).astype(np.int32)
        lens = rng.randint(0, T + 1, size=N)
        for n in range(N):
            tok[:T - lens[n], n] = 0
        W1 = (f32(rng, D + H, 4 * H) / np.sqrt(D + H)).astype(np.float32); b1 = f32(rng, 4 * H) * 0.1
        W2 = (f32(rng, 2 * H, 4 * H) / np.sqrt(2 * H)).astype(np.float32); b2 = f32(rng, 4 * H) * 0.1
        x = emb[tok]
        d = np.float64
        h1, c1, g1 = vo.lstm_forward(x.astype(d), W1.astype(d), b1.astype(d), tok)
        h2, c2, g2 = vo.lstm_forward(h1, W2.astype(d), b2.astype(d), tok)
        dlast = f32(rng, N, H)
        dx2, dW2, db2, _, _, da2 = vo.lstm_backward(h1, W2.astype(d), g2, h2, c2, dh_last=dlast.astype(d), return_da=True)
        dx1, dW1, db1, _, _, da1 = vo.lstm_backward(x.astype(d), W1.astype(d), g1, h1, c1, dh_seq=dx2, return_da=True)
        t = lambda *s: torch.empty(*s, device="cuda")
        dev_st = dict(T=T, N=N, tok_mask=dev(tok), Wh1=dev(W1[D:]), Wx2=dev(W2[:H]), b2=dev(b2), Wh2=dev(W2[H:]),
                      gates1=dev((x.reshape(T * N, D).astype(d) @ W1[:D].astype(d) + b1).astype(np.float32).reshape(T, N, 4 * H)),
                      h1=t(T, N, H), c1=t(T, N, H), gates2=t(T, N, 4 * H), h2=t(T, N, H), c2=t(T, N, H))
        stacks.append((dev_st, dict(h1=h1, h2=h2, c2=c2, g1=g1, g2=g2, da1=da1, da2=da2, dlast=dlast)))
    fwd([s for s, _ in stacks], H)
    torch.cuda.synchronize()
    for s, r in stacks:
        assert relerr(s['h1'], r['h1']) < tol_f and relerr(s['h2'], r['h2']) < tol_f
        assert relerr(s['gates1'], r['g1']) < tol_f and relerr(s['gates2'], r['g2']) < tol_f and relerr(s['c2'], r['c2']) < tol_f
        assert arith == 'fp32' or relerr(s['h2'], r['h2']) > 1e-5       # the switch really changes the arithmetic
    bw = []
    for s, r in stacks:
        T, N = s['T'], s['N']
        bw.append(dict(T=T, N=N, Wh1=s['Wh1'], Wx2=s['Wx2'], Wh2=s['Wh2'], gates1=s['gates1'], c1=s['c1'],
                       gates2=s['gates2'], c2=s['c2'], dh_last2=dev(r['dlast']), dh1_seq=torch.empty(T, N, H, device="cuda"),
                       dc1=torch.empty(N, H, device="cuda"), dc2=torch.empty(N, H, device="cuda")))
    bwd(bw, H)
    torch.cuda.synchronize()
    for s, r in stacks:
        assert relerr(s['gates2'], r['da2']) < tol_b and relerr(s['gates1'], r['da1']) < tol_b


def test_bf16_compact_state_saturated_gates(ops):
    """configs[4] compact bf16 state (csrc/lstm.hip, EpiLstmFwdT<.., C16>): a sigmoid gate s > 1/2 is stored as s - 1.  A gate that
    saturates to exactly 1.0f (pre-activation above ~16.6) must not be stored as +0 -- the backward pass would read s = 0 and cut the
    cell gradient through a saturated forget gate.  Forget-gate pre-activations of +20 / +40 on most rows: dc0 and da
    against the fp64 recurrence on the same bf16-rounded table and weights."""
    import ctypes as C
    from visdial_amd import _lib
    lib = _lib.load()
    fwd = getattr(lib, '_Z19vd_lstm_forward_c16PKtlPKiPKfPtS5_PfS6_iiiP12ihipStream_t')
    bwd = getattr(lib, '_Z20vd_lstm_backward_c16PKfPtS0_S0_PfiiiP12ihipStream_t')
    p = C.c_void_p
    fwd.argtypes = [p, C.c_long, p, p, p, p, p, p, C.c_int, C.c_int, C.c_int, p]
    bwd.argtypes = [p, p, p, p, p, C.c_int, C.c_int, C.c_int, p]
    T, N, H, V = 4, 2048, 128, 50
    rng = np.random.RandomState(11)
    Wh = (f32(rng, H, 4 * H) / np.sqrt(H) * 0.5).astype(np.float32)
    tab = f32(rng, V + 1, 4 * H) * 0.5
    tab[1:40, H:2 * H] += 20.0           # forget gate saturates to exactly 1.0f in fp32 ...
    tab[40:, H:2 * H] += 40.0
    tab[1:20, 0:H] += 18.0               # ... and so do some input / output gates
    tab[20:40, 2 * H:3 * H] += 25.0
    tab16 = dev(tab).to(torch.bfloat16)
    Wh16 = dev(Wh).to(torch.bfloat16)
    tok = rng.randint(1, V + 1, size=(T, N)).astype(np.int32)
    dh_last = f32(rng, N, H)
    gates16 = torch.empty(T, N, 4 * H, device='cuda', dtype=torch.bfloat16)
    h16 = torch.empty(T, N, H, device='cuda', dtype=torch.bfloat16)
    h_last = torch.empty(N, H, device='cuda')
    c = torch.empty(T, N, H, device='cuda')
    dc = torch.empty(N, H, device='cuda')
    Whd, tokd, dhd = dev(Wh), dev(tok), dev(dh_last)
    stream = torch.cuda.current_stream().cuda_stream
    assert fwd(tab16.data_ptr(), 4 * H, tokd.data_ptr(), Whd.data_ptr(), gates16.data_ptr(), h16.data_ptr(), h_last.data_ptr(),
               c.data_ptr(), T, N, H, stream) == 0
    torch.cuda.synchronize()
    # fp64 recurrence on the operands the kernel multiplies: bf16 table rows, bf16 weights (h is rounded to bf16 between steps: inside the bound)
    xw = tab16.float().cpu().numpy().astype(np.float64)[tok]                       # [T, N, 4H] = the gathered projection rows
    W = np.concatenate([np.eye(4 * H), Wh16.float().cpu().numpy().astype(np.float64)], 0)    # x = the projection itself (Wx = I, b = 0)
    h_ref, c_ref, g_ref = vo.lstm_forward(xw, W, np.zeros(4 * H))
    assert (g_ref[:, :, H:2 * H].astype(np.float32) == 1.0).mean() > 0.5           # the case under test is present
    assert relerr(c, c_ref) < 1e-2 and relerr(h_last, h_ref[-1]) < 1e-2
    assert bwd(Whd.data_ptr(), gates16.data_ptr(), c.data_ptr(), dhd.data_ptr(), dc.data_ptr(), T, N, H, stream) == 0
    torch.cuda.synchronize()
    _, _, _, _, dc0_ref, da_ref = vo.lstm_backward(xw, W, g_ref, h_ref, c_ref, dh_last=dh_last.astype(np.float64), return_da=True)
    # with the +0 encoding dc0 lost every path through a saturated forget gate (rel-L2 ~ 1); the bf16 bound of the config is 2e-2
    assert relerr(dc, dc0_ref) < 2e-2, relerr(dc, dc0_ref)
    assert relerr(gates16.float(), da_ref) < 2e-2, relerr(gates16.float(), da_ref)


@pytest.mark.parametrize("M,N,K", [(128, 256, 4096 + 7), (256, 512, 8192 + 32 * 5 + 9), (512, 2048, 3000 * 19), (384, 256, 9000)])
def test_bf16_weight_gradient_contraction(ops, M, N, K):
    """configs[4] weight-gradient contraction dWh += h16^T * da16 over the compact bf16 state (csrc/gemm_ops.hip `vd_gemm_tn_acc_bf16`):
    LDS-DMA tiles + transpose reads, 128 x 128 tiles (any M, N % 128 == 0) or 256 x 256 tiles with eight waves (M, N % 256 == 0 and
    K >= 8192), split-K partial sums by atomics, the last < 32 rows by a tail kernel.  bf16 products are exact in fp32, so the only
    difference to an fp32 matmul of the same bf16 values is the summation order: 1e-5 relative.  ACCUMULATES into C."""
    import ctypes as C
    from visdial_amd import _lib
    lib = _lib.load()
    tn = getattr(lib, '_Z19vd_gemm_tn_acc_bf16PKtS0_PfliiiP12ihipStream_t')
    p = C.c_void_p
    tn.argtypes = [p, p, p, C.c_long, C.c_int, C.c_int, C.c_int, p]
    g = torch.Generator(device='cuda').manual_seed(5)
    a16 = (torch.randn(K, M, device='cuda', generator=g) * 0.3).to(torch.bfloat16)
    b16 = (torch.randn(K, N, device='cuda', generator=g) * 0.05).to(torch.bfloat16)
    c0 = torch.randn(M, N, device='cuda', generator=g)
    c = c0.clone()
    assert tn(a16.data_ptr(), b16.data_ptr(), c.data_ptr(), N, M, N, K, torch.cuda.current_stream().cuda_stream) == 0, lib.vd_last_error()
    torch.cuda.synchronize()
    ref = c0.double() + a16.double().t() @ b16.double()
    err = ((c.double() - ref).norm() / ref.norm()).item()
    assert err < 1e-5, err


def test_bf16_segment_rowsum(ops):
    """configs[4] projection-table gradient: out[token] += sum of the bf16 rows of that token, rows visited in token order (csrc/elementwise.hip
    `vd_segment_rowsum_acc_bf16`: 16-byte loads, fp32 sums, a run of one token that spans two chunks flushed by float atomics).  Against
    index_add of the same bf16 values in fp64; ragged chunk at the end, a long run (the most frequent token) and absent tokens."""
    import ctypes as C
    from visdial_amd import _lib
    lib = _lib.load()
    fn = getattr(lib, '_Z26vd_segment_rowsum_acc_bf16PKtlPKiS2_liPflP12ihipStream_t')
    p = C.c_void_p
    fn.argtypes = [p, C.c_long, p, p, C.c_long, C.c_int, p, C.c_long, p]
    rng = np.random.RandomState(3)
    n, ncol, V1 = 5000 + 37, 512, 300
    tok = rng.randint(1, V1, size=n).astype(np.int32)
    tok[rng.rand(n) < 0.3] = 7                                   # one long run
    tok[(tok > 100) & (tok < 120)] = 99                           # tokens 101..119 never occur
    perm = np.argsort(tok, kind='stable').astype(np.int32)
    x16 = (torch.randn(n, ncol, device='cuda') * 0.1).to(torch.bfloat16)
    out0 = torch.randn(V1, ncol, device='cuda')
    out = out0.clone()
    tok_d, perm_d = dev(tok), dev(perm)
    assert fn(x16.data_ptr(), ncol, tok_d.data_ptr(), perm_d.data_ptr(), n, ncol, out.data_ptr(), ncol, torch.cuda.current_stream().cuda_stream) == 0
    torch.cuda.synchronize()
    ref = out0.double().index_add(0, torch.from_numpy(tok.astype(np.int64)).cuda(), x16.double())
    assert float((out.double() - ref).norm() / ref.norm()) < 1e-6


@pytest.mark.parametrize("M,N,K", [(512, 2048, 8000), (256, 128, 1024 + 16 * 3 + 5), (512, 256, 4000)])
def test_bf16_contraction_of_fp32_rows(ops, M, N, K):
    """The encoder's dense weight gradients in a bf16 pass (csrc/gemm_ops.hip `vd_gemm_tn_acc(flags = VD_FLAG_BF16)` on fp32 operands with
    no bf16 shadow: M % 256 == 0, N % 128 == 0, K >= 1024 take gemm_split_tn_kernel<1> -- LDS-DMA tiles of the fp32 rows, both operands
    rounded to bf16 (RNE) in registers, fp32 accumulation; the last K % 16 rows through the staging kernel).  Exact against the fp64 product of
    the bf16-ROUNDED operands up to summation order (1e-5); against the fp32 operands it is the bf16 pass's error (a few 1e-3)."""
    g = torch.Generator(device='cuda').manual_seed(9)
    a = torch.tanh(torch.randn(K, M, device='cuda', generator=g))
    b = torch.randn(K, N, device='cuda', generator=g) * 0.01
    c0 = torch.randn(M, N, device='cuda', generator=g) * 0.1
    c = c0.clone()
    ops.gemm_tn_acc(a, b, c, M=M, N=N, K=K, flags=ops.FLAG_BF16)
    torch.cuda.synchronize()
    ref16 = c0.double() + a.to(torch.bfloat16).double().t() @ b.to(torch.bfloat16).double()
    ref32 = c0.double() + a.double().t() @ b.double()
    e16 = float((c.double() - ref16).norm() / ref16.norm())
    e32 = float((c.double() - ref32).norm() / ref32.norm())
    assert e16 < 1e-5, e16
    assert 1e-5 < e32 < 1e-2, e32
