"""Thin torch-tensor wrappers over the C ABI (include/visdial_hip.h).

torch is plumbing only here: it owns device memory and the current HIP stream; every
computation is a call into libvisdial_hip.so.  All tensors must be contiguous CUDA(HIP)
tensors of the documented dtype (float32 / int32 / uint8).
"""
import torch

from . import _lib

call = _lib.call


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _p(t, dtype=None):
    if t is None:
        return None
    assert t.is_cuda, "device tensor expected"
    assert t.is_contiguous(), "contiguous tensor expected"
    if dtype is not None:
        assert t.dtype == dtype, "expected %s got %s" % (dtype, t.dtype)
    return t.data_ptr()


F32, I32, U8 = torch.float32, torch.int32, torch.uint8

# ---------------------------------------------------------------- live kernel-family timing (bench.py)
# HIP events recorded on the launch stream (torch's current stream) around a group of launches.
PROFILE = None   # None = off; dict tag -> list of (start_event, end_event, n_launches)


def prof_begin(tag):
    if PROFILE is None:
        return None
    e = torch.cuda.Event(enable_timing=True)
    e.record()
    return e


def prof_end(tag, start, n_launches=1):
    if PROFILE is None or start is None:
        return
    e = torch.cuda.Event(enable_timing=True)
    e.record()
    PROFILE.setdefault(tag, []).append((start, e, n_launches))


def prof_summary():
    """tag -> (total_ms, n_launches).  Call after torch.cuda.synchronize()."""
    out = {}
    for tag, evs in (PROFILE or {}).items():
        ms = sum(a.elapsed_time(b) for a, b, _ in evs)
        out[tag] = (ms, sum(n for _, _, n in evs))
    return out


# ---------------------------------------------------------------- dense contractions
def gemm_nt(A, W, C_out, bias=None, act=0, accumulate=False, M=None, N=None, K=None, lda=None, ldw=None, ldc=None):
    """C[MxN] (+)= act(A[MxK] @ W[NxK].T + bias)"""
    M = A.shape[0] if M is None else M
    K = A.shape[1] if K is None else K
    N = W.shape[0] if N is None else N
    call("vd_gemm_nt", _p(A, F32), lda or A.stride(0), _p(W, F32), ldw or W.stride(0), _p(bias, F32), _p(C_out, F32),
         ldc or C_out.stride(0), M, N, K, act, int(accumulate), _stream())
    return C_out


def gemm_nn(A, B, C_out, bias=None, accumulate=False, M=None, N=None, K=None, lda=None, ldb=None, ldc=None):
    """C[MxN] (+)= A[MxK] @ B[KxN] + bias"""
    M = A.shape[0] if M is None else M
    K = A.shape[1] if K is None else K
    N = B.shape[1] if N is None else N
    call("vd_gemm_nn", _p(A, F32), lda or A.stride(0), _p(B, F32), ldb or B.stride(0), _p(bias, F32), _p(C_out, F32),
         ldc or C_out.stride(0), M, N, K, int(accumulate), _stream())
    return C_out


FLAG_BF16 = 1      # VD_FLAG_BF16: bf16 operands / fp32 accumulation (opt-in, BASELINE configs[4])
FLAG_SPLIT9, FLAG_SPLIT6, FLAG_SPLIT3 = 2, 4, 8   # exact three-way bf16 split of both operands: 9 products = fp32-grade (opt-in)
PRECISION_FLAGS = {'fp32': 0, 'bf16': FLAG_BF16, 'split9': FLAG_SPLIT9, 'split6': FLAG_SPLIT6, 'split3': FLAG_SPLIT3}
PRECISION_CODES = {'fp32': 0, 'bf16': 1, 'split9': 9, 'split6': 6, 'split3': 3}        # vd_model_params.lstmBf16


def gemm_tn_acc(A, B, C_acc, M=None, N=None, K=None, lda=None, ldb=None, ldc=None, flags=0):
    """C[MxN] += A[KxM].T @ B[KxN]"""
    K = A.shape[0] if K is None else K
    M = A.shape[1] if M is None else M
    N = B.shape[1] if N is None else N
    call("vd_gemm_tn_acc", _p(A, F32), lda or A.stride(0), _p(B, F32), ldb or B.stride(0), _p(C_acc, F32),
         ldc or C_acc.stride(0), M, N, K, int(flags), _stream())
    return C_acc


def gemm_tn_rows_acc(A, a_rows, B, b_rows, C_acc, M=None, N=None, lda=None, ldb=None, ldc=None):
    """C[MxN] += sum_k A[a_rows[k], :M].T @ B[b_rows[k], :N]  (K = len(a_rows) = len(b_rows) row pairs)"""
    K = a_rows.numel()
    assert b_rows.numel() == K
    M = A.shape[1] if M is None else M
    N = B.shape[1] if N is None else N
    call("vd_gemm_tn_rows_acc", _p(A, F32), lda or A.stride(0), _p(a_rows, I32), _p(B, F32), ldb or B.stride(0),
         _p(b_rows, I32), _p(C_acc, F32), ldc or C_acc.stride(0), M, N, K, _stream())
    return C_acc


def colsum_acc(X, out, M=None, N=None, ld=None):
    M = X.shape[0] if M is None else M
    N = X.shape[1] if N is None else N
    call("vd_colsum_acc", _p(X, F32), ld or X.stride(0), M, N, _p(out, F32), _stream())
    return out


# ---------------------------------------------------------------- LSTM
def lstm_forward(xproj, Wh, gates, h, c, T, N, H, x_tstride, x_ld, tok_gather=None, tok_mask=None, h0=None, c0=None,
                 flags=0):
    call("vd_lstm_forward", _p(xproj, F32), x_tstride, x_ld, _p(tok_gather, I32), _p(tok_mask, I32), _p(Wh, F32),
         _p(h0, F32), _p(c0, F32), _p(gates, F32), _p(h, F32), _p(c, F32), T, N, H, int(flags), _stream())


def lstm_backward(Wh, gates, c, dc_work, T, N, H, c0=None, dh_seq=None, dh_last=None, dc_last=None, dh0=None,
                  flags=0, h_seq=None, dWh=None):
    """h_seq + dWh: also accumulate dWh += sum_t h_{t-1}^T da_t (overlapped with the recurrence)."""
    call("vd_lstm_backward", _p(Wh, F32), _p(gates, F32), _p(c, F32), _p(c0, F32), _p(dh_seq, F32), _p(dh_last, F32),
         _p(dc_last, F32), _p(dc_work, F32), _p(dh0, F32), _p(h_seq, F32), _p(dWh, F32), T, N, H, int(flags),
         _stream())


# ---------------------------------------------------------------- embedding / dropout / glue
def embed_gather(emb, tok, out, mask=None, scale=1.0):
    rows = tok.numel()
    call("vd_embed_gather", _p(emb, F32), _p(tok, I32), _p(mask, U8), _p(out, F32), rows, emb.shape[1], float(scale),
         _stream())
    return out


def embed_scatter_acc(demb, tok, dx, mask=None, scale=1.0):
    rows = tok.numel()
    call("vd_embed_scatter_acc", _p(demb, F32), _p(tok, I32), _p(mask, U8), _p(dx, F32), rows, demb.shape[1],
         float(scale), _stream())


def token_sort(tok, V, offset, work, perm):
    call("vd_token_sort", _p(tok, I32), tok.numel(), V, _p(offset, I32), _p(work, I32), _p(perm, I32), _stream())


def segment_rowsum_acc(X, tok, perm, out, ncol=None):
    ncol = X.shape[1] if ncol is None else ncol
    call("vd_segment_rowsum_acc", _p(X, F32), X.stride(0), _p(tok, I32), _p(perm, I32), tok.numel(), ncol,
         _p(out, F32), out.stride(0), _stream())


def dropout_mask(mask, seed, p):
    call("vd_dropout_mask", _p(mask, U8), mask.numel(), int(seed) & 0xFFFFFFFFFFFFFFFF, float(p), _stream())
    return mask


def dropout_apply(x, mask, y, scale):
    call("vd_dropout_apply", _p(x, F32), _p(mask, U8), _p(y, F32), x.numel(), float(scale), _stream())
    return y


def tanh_backward(dy, y, dx):
    call("vd_tanh_backward", _p(dy, F32), _p(y, F32), _p(dx, F32), y.numel(), _stream())
    return dx


def axpby(a, b, c, alpha=1.0, beta=1.0):
    call("vd_axpby", _p(a, F32), _p(b, F32), _p(c, F32), a.numel(), float(alpha), float(beta), _stream())
    return c


# ---------------------------------------------------------------- attention
def mn_attention_forward(Q, Hm, mask, P, hAtt, B, R, H):
    call("vd_mn_attention_forward", _p(Q, F32), _p(Hm, F32), _p(mask, U8), _p(P, F32), _p(hAtt, F32), B, R, H,
         _stream())


def mn_attention_backward(Q, Hm, P, dhAtt, dQ, dHm, B, R, H):
    call("vd_mn_attention_backward", _p(Q, F32), _p(Hm, F32), _p(P, F32), _p(dhAtt, F32), _p(dQ, F32), _p(dHm, F32),
         B, R, H, _stream())


def img_common_forward(pre, mask1, Wc, bc, qc, mask2, iqc, N, R, S2, H, Kc, scale):
    call("vd_img_common_forward", _p(pre, F32), _p(mask1, U8), _p(Wc, F32), _p(bc, F32), _p(qc, F32), _p(mask2, U8),
         _p(iqc, F32), N, R, S2, H, Kc, float(scale), _stream())


def img_att_forward(iqc, wa, ba, pre, mask1, u0, p, u1, N, R, S2, H, Kc, scale):
    call("vd_img_att_forward", _p(iqc, F32), _p(wa, F32), _p(ba, F32), _p(pre, F32), _p(mask1, U8), _p(u0, F32),
         _p(p, F32), _p(u1, F32), N, R, S2, H, Kc, float(scale), _stream())


def img_att_backward(iqc_dz, wa, pre, mask1, mask2, p, datt, dwa, dba, dqc, work, N, R, S2, H, Kc, scale):
    """work: [N x S2] fp32 scratch (softmax-backward scores)."""
    assert work.numel() >= N * S2
    call("vd_img_att_backward", _p(iqc_dz, F32), _p(wa, F32), _p(pre, F32), _p(mask1, U8), _p(mask2, U8), _p(p, F32),
         _p(datt, F32), _p(dwa, F32), _p(dba, F32), _p(dqc, F32), _p(work, F32), N, R, S2, H, Kc, float(scale),
         _stream())


def img_tr_backward(dz, Wc, p, datt, mask1, dpre, N, R, S2, H, Kc, scale):
    call("vd_img_tr_backward", _p(dz, F32), _p(Wc, F32), _p(p, F32), _p(datt, F32), _p(mask1, U8), _p(dpre, F32), N, R,
         S2, H, Kc, float(scale), _stream())


def img_common_wgrad(dz, pre, mask1, dWc, N, R, S2, H, Kc, scale):
    call("vd_img_common_wgrad", _p(dz, F32), _p(pre, F32), _p(mask1, U8), _p(dWc, F32), N, R, S2, H, Kc, float(scale),
         _stream())


# ---------------------------------------------------------------- head / optimiser
def score_ce(optH, enc, scores, N, O, H, gt=None, loss_rows=None, dOptH=None, dEnc=None, gscale=1.0):
    call("vd_score_ce", _p(optH, F32), _p(enc, F32), _p(gt, I32), _p(scores, F32), _p(loss_rows, F32), _p(dOptH, F32),
         _p(dEnc, F32), N, O, H, float(gscale), _stream())


def ranks(scores, out, N, O):
    call("vd_ranks", _p(scores, F32), _p(out, I32), N, O, _stream())
    return out


def clamp_adam(w, g, m, v, step, gscale=1.0, clip=5.0, beta1=0.9, beta2=0.999, eps=1e-8):
    call("vd_clamp_adam", _p(w, F32), _p(g, F32), _p(m, F32), _p(v, F32), w.numel(), float(gscale), float(clip),
         float(beta1), float(beta2), float(eps), float(step), _stream())


# ---------------------------------------------------------------- widening: gen head, MaskTime, column copies
def logsoftmax_nll(logits, V, tok_in, target, loss_rows, write_grad=True):
    rows = tok_in.numel()
    call("vd_logsoftmax_nll", _p(logits, F32), logits.stride(0), rows, V, _p(tok_in, I32), _p(target, I32),
         _p(loss_rows, F32), int(write_grad), _stream())


def mask_time_forward(feat, tok, out, T, N, D):
    call("vd_mask_time_forward", _p(feat, F32), _p(tok, I32), _p(out, F32), T, N, D, _stream())
    return out


def mask_time_backward(dout, tok, dfeat, T, N, D):
    call("vd_mask_time_backward", _p(dout, F32), _p(tok, I32), _p(dfeat, F32), T, N, D, _stream())
    return dfeat


def copy_2d(dst, dst_ld, src, src_ld, rows, cols, dst_off=0, src_off=0):
    """copy a [rows x cols] column block; offsets are in floats from the tensor base"""
    assert dst.is_cuda and src.is_cuda and dst.dtype == F32 and src.dtype == F32
    call("vd_copy_2d", dst.data_ptr() + 4 * dst_off, dst_ld, src.data_ptr() + 4 * src_off, src_ld, rows, cols, _stream())


def _lstm2_fwd_array(stacks):
    arr = (_lib.Lstm2Fwd * len(stacks))()
    for a, s in zip(arr, stacks):
        a.T, a.N = s['T'], s['N']
        a.tok_mask = _p(s.get('tok_mask'), I32)
        for k in ('Wh1', 'Wx2', 'b2', 'Wh2', 'gates1', 'h1', 'c1', 'gates2', 'h2', 'c2'):
            setattr(a, k, _p(s[k], F32))
        na = s.get('nact')            # host numpy int32[T] (kept alive by the caller's dict) or None
        a.nact = na.ctypes.data if na is not None else None
    return arr


def _lstm2_bwd_array(stacks):
    arr = (_lib.Lstm2Bwd * len(stacks))()
    for a, s in zip(arr, stacks):
        a.T, a.N = s['T'], s['N']
        for k in ('Wh1', 'Wx2', 'Wh2', 'gates1', 'c1', 'gates2', 'c2', 'dh_last2', 'dh1_seq', 'dc1', 'dc2'):
            setattr(a, k, _p(s[k], F32))
        na = s.get('nact')
        a.nact = na.ctypes.data if na is not None else None
    return arr


def lstm2_forward(stacks, H):
    """stacks: list of dicts with the vd_lstm2_fwd_t fields (tensors)."""
    import ctypes
    call("vd_lstm2_forward", ctypes.cast(_lstm2_fwd_array(stacks), ctypes.c_void_p), len(stacks), H, _stream())


def lstm2_backward(stacks, H):
    import ctypes
    call("vd_lstm2_backward", ctypes.cast(_lstm2_bwd_array(stacks), ctypes.c_void_p), len(stacks), H, _stream())


def lstm2_pass(stacks, H, flags, backward=False):
    """The wavefront with a pass's arithmetic (flags = FLAG_BF16: the encoder ticks of a bf16 pass of the model-level runtime):
    `vd_lstm2_forward_flags / vd_lstm2_backward_flags` (the C ABI's vd_lstm2_forward / _backward are these with flags = 0)."""
    import ctypes
    arr = _lstm2_bwd_array(stacks) if backward else _lstm2_fwd_array(stacks)
    call("vd_lstm2_backward_flags" if backward else "vd_lstm2_forward_flags", ctypes.cast(arr, ctypes.c_void_p), len(stacks), H, int(flags),
         _stream())


def hrea_attention_forward(sq, sh, Hm, P, att, B, R, H):
    call("vd_hrea_attention_forward", _p(sq, F32), _p(sh, F32), _p(Hm, F32), _p(P, F32), _p(att, F32), B, R, H, _stream())


def hrea_attention_backward(Hm, P, datt, dsq, dsh, dHm, B, R, H):
    call("vd_hrea_attention_backward", _p(Hm, F32), _p(P, F32), _p(datt, F32), _p(dsq, F32), _p(dsh, F32),
         _p(dHm, F32), B, R, H, _stream())


def rowdot_forward(x, w, bias, out, N, H):
    call("vd_rowdot_forward", _p(x, F32), _p(w, F32), _p(bias, F32), _p(out, F32), N, H, _stream())
    return out


def rowdot_backward(x, w, dout, dw, db, dx, N, H):
    call("vd_rowdot_backward", _p(x, F32), _p(w, F32), _p(dout, F32), _p(dw, F32), _p(db, F32), _p(dx, F32), N, H,
         _stream())
    return dx


def zero_inactive_rows(buf, nact_dev, T, N, ncols):
    """buf [T, N, ncols] contiguous"""
    call("vd_zero_inactive_rows", _p(buf, F32), N * ncols, ncols, ncols, _p(nact_dev, I32), T, N, _stream())


def log_softmax_rows(x, V):
    call("vd_log_softmax_rows", _p(x, F32), x.stride(0), x.shape[0], V, _stream())
    return x


def zero(t):
    """t[...] = 0 through vd_memset (hipMemsetAsync on the current stream); contiguous tensors only"""
    call("vd_memset", _p(t), 0, t.numel() * t.element_size(), _stream())
    return t
