"""Pins the numpy oracle (oracle/visdial_oracle.py) with an INDEPENDENT torch-autograd restatement of the
same forward graph and with finite differences.  (The reference has no golden vectors -- SURVEY.md 8c.)"""
import numpy as np
import pytest
import torch

from conftest import small_params
from oracle import visdial_oracle as vo
from visdial_amd.dataloader import SyntheticDataloader


def make_drop(p, batch, rng):
    B, R, Tq = batch['ques_fwd'].shape
    Th = batch['hist'].shape[2]
    N, H, E = B * R, p['rnnHiddenSize'], p['embedSize']
    S2, K = p['imgSpatialSize'] ** 2, p['commonEmbeddingSize']
    shp = dict(q_emb=(Tq, N, E), h_emb=(Th, N, E), hatt=(N, H), img_tr=(N, S2, H), iqc=(N, S2, K), u=(N, H))
    for sfx in vo.hop_suffixes(p)[1:]:
        shp['iqc' + sfx] = (N, S2, K)
    return {k: (rng.rand(*s) > 0.5).astype(np.float64) for k, s in shp.items()}


def torch_lstm(x, W, b, tok, H):
    """independent restatement with torch ops + autograd"""
    T, N, D = x.shape
    h = torch.zeros(N, H, dtype=x.dtype)
    c = torch.zeros(N, H, dtype=x.dtype)
    hs = []
    for t in range(T):
        a = torch.addmm(b, torch.cat([x[t], h], 1), W)
        i, f, o, g = a[:, :H].sigmoid(), a[:, H:2 * H].sigmoid(), a[:, 2 * H:3 * H].sigmoid(), a[:, 3 * H:].tanh()
        c = f * c + i * g
        h = o * c.tanh()
        if tok is not None:
            keep = (tok[t] != 0).to(x.dtype)[:, None]
            h, c = h * keep, c * keep
        hs.append(h)
    return torch.stack(hs)


def torch_forward(P, p, batch, drop):
    F = torch.nn.functional
    B, R, Tq = batch['ques_fwd'].shape
    N, H = B * R, p['rnnHiddenSize']
    S2 = p['imgSpatialSize'] ** 2
    dm = {k: torch.from_numpy(v) for k, v in drop.items()} if drop else None
    dr = (lambda x, k: x * dm[k] * 2.0) if drop else (lambda x, k: x)
    emb = P['embed'] * torch.cat([torch.zeros(1, 1, dtype=torch.float64), torch.ones(P['embed'].shape[0] - 1, 1, dtype=torch.float64)])
    qtok = torch.from_numpy(batch['ques_fwd'].reshape(N, -1).T.astype(np.int64))
    htok = torch.from_numpy(batch['hist'].reshape(N, -1).T.astype(np.int64))
    qx, hx = dr(emb[qtok], 'q_emb'), dr(emb[htok], 'h_emb')
    h3 = torch_lstm(torch_lstm(hx, P['hist1.W'], P['hist1.b'], htok, H), P['hist2.W'], P['hist2.b'], htok, H)[-1]
    q3 = torch_lstm(torch_lstm(qx, P['ques1.W'], P['ques1.b'], qtok, H), P['ques2.W'], P['ques2.b'], qtok, H)[-1]
    qv, hv = q3.view(B, R, H), h3.view(B, R, H)
    s = torch.bmm(qv, hv.transpose(1, 2))
    mask = torch.triu(torch.ones(R, R, dtype=torch.bool), 1)
    s = s.masked_fill(mask[None], -9999999.0)
    hatt = torch.bmm(torch.softmax(s, -1), hv).reshape(N, H)
    hattTr = torch.tanh(F.linear(dr(hatt, 'hatt'), P['mn1.W'], P['mn1.b']))
    qh2 = torch.tanh(F.linear(hattTr + q3, P['mn2.W'], P['mn2.b']))
    img = torch.from_numpy(batch['img_feat'].astype(np.float64)).reshape(B, S2, -1)
    img = img[:, None].expand(B, R, S2, img.shape[-1]).reshape(N, S2, -1)      # model.lua:262-265
    img_tr = dr(torch.tanh(F.linear(img, P['img_proj.W'], P['img_proj.b'])), 'img_tr')
    u = qh2
    for sfx in vo.hop_suffixes(p):                                             # mn-att:82-104, one hop per iteration
        ic = F.linear(img_tr, P['img_common%s.W' % sfx], P['img_common%s.b' % sfx])
        qc = F.linear(u, P['ques_common%s.W' % sfx], P['ques_common%s.b' % sfx])
        iqc = dr(torch.tanh(ic + qc[:, None, :]), 'iqc' + sfx)
        patt = torch.softmax(F.linear(iqc, P['att%s.W' % sfx], P['att%s.b' % sfx]).squeeze(-1), 1)
        u = torch.bmm(patt[:, None, :], img_tr).squeeze(1) + u
    enc = torch.tanh(F.linear(dr(u, 'u'), P['out.W'], P['out.b']))
    opt = batch['options']
    O = opt.shape[1]
    otok = torch.from_numpy(opt.reshape(N * O, -1).T.astype(np.int64))
    oh = torch_lstm(emb[otok], P['opt.W'], P['opt.b'], None, H)[-1].view(N, O, H)
    scores = torch.bmm(oh, enc[:, :, None]).squeeze(-1)
    loss = F.cross_entropy(scores, torch.from_numpy(batch['answer_ind'].astype(np.int64)) - 1)
    return loss, scores


@pytest.mark.parametrize("hops", [1, 3])
@pytest.mark.parametrize("use_drop", [False, True])
def test_oracle_matches_torch_autograd(use_drop, hops):
    p = small_params(numAttentionLayers=hops)
    dl = SyntheticDataloader(p, seed=7)
    batch = dl.getTrainBatch(p)
    P = vo.init_params(p['encoder'], p['decoder'], p, seed=3)
    rng = np.random.RandomState(11)
    drop = make_drop(p, batch, rng) if use_drop else None
    r = vo.forward_backward(p['encoder'], p['decoder'], P, p, batch, drop)
    Pt = {k: torch.from_numpy(v.copy()).requires_grad_(True) for k, v in P.items()}
    loss, scores = torch_forward(Pt, p, batch, drop)
    loss.backward()
    assert abs(loss.item() - r['loss']) < 1e-10
    np.testing.assert_allclose(scores.detach().numpy(), r['scores'], rtol=1e-9, atol=1e-10)
    for k in P:
        g_t = Pt[k].grad.numpy()
        g_o = r['grads'][k]
        if k == 'embed':
            # the pad row: torch sees emb*0 (no grad); the reference's dense gradWeight accumulates pad
            # gradients into row 0 -- compare the real rows only.
            g_t, g_o = g_t[1:], g_o[1:]
        np.testing.assert_allclose(g_o, g_t, rtol=1e-7, atol=1e-10, err_msg=k)


def test_oracle_finite_difference():
    p = small_params(batchSize=1, maxQuesCount=3, numOptions=4)
    dl = SyntheticDataloader(p, seed=5)
    batch = dl.getTrainBatch(p)
    P = vo.init_params(p['encoder'], p['decoder'], p, seed=2)
    r = vo.forward_backward(p['encoder'], p['decoder'], P, p, batch, None)
    rng = np.random.RandomState(0)
    eps = 1e-6
    for name in ['hist1.W', 'ques2.b', 'mn1.W', 'img_proj.W', 'img_common.b', 'att.W', 'out.W', 'opt.W', 'embed']:
        w = P[name]
        for _ in range(3):
            idx = tuple(rng.randint(0, s) for s in w.shape)
            if name == 'embed' and idx[0] == 0:
                continue
            old = w[idx]
            w[idx] = old + eps
            lp = vo.forward_backward(p['encoder'], p['decoder'], P, p, batch, None, only_forward=True)['loss']
            w[idx] = old - eps
            lm = vo.forward_backward(p['encoder'], p['decoder'], P, p, batch, None, only_forward=True)['loss']
            w[idx] = old
            fd = (lp - lm) / (2 * eps)
            assert abs(fd - r['grads'][name][idx]) < 1e-6 * max(1.0, abs(fd)), (name, idx, fd, r['grads'][name][idx])


def test_ranks_and_metrics():
    s = np.array([[0.1, 0.9, 0.5, 0.9], [3.0, 2.0, 1.0, 0.0]])
    r = vo.compute_ranks(s)
    assert r.tolist() == [[4, 1, 3, 2], [1, 2, 3, 4]]
    assert vo.compute_ranks(s, np.array([2, 0])).tolist() == [3, 1]
    m = vo.process_ranks(np.array([[1, 2], [6, 11]]))
    assert m['r@1'] == 0.25 and m['r@5'] == 0.5 and m['r@10'] == 0.75
    assert abs(m['meanRR'] - (1 + 0.5 + 1 / 6 + 1 / 11) / 4) < 1e-12


def test_adam_matches_reference_formula():
    rng = np.random.RandomState(0)
    w = rng.randn(10); g = rng.randn(10) * 10
    st = {}
    w1, gc = vo.clamp_adam(w, g, st, 1e-3)
    gc2 = np.clip(g, -5, 5)
    m = 0.1 * gc2; v = 0.001 * gc2 * gc2
    step = 1e-3 * np.sqrt(1 - 0.999) / (1 - 0.9)
    np.testing.assert_allclose(w1, w - step * m / (np.sqrt(v) + 1e-8), rtol=1e-12)


# ----------------------------------------------------------------------------------------------------
# widening rows a10 / a11: lf-ques, lf-ques-im-hist, hre-ques-im-hist encoders and the gen decoder
def torch_lstm2(x, W, b, tok, H, h0=None, c0=None):
    T, N, D = x.shape
    h = torch.zeros(N, H, dtype=x.dtype) if h0 is None else h0
    c = torch.zeros(N, H, dtype=x.dtype) if c0 is None else c0
    hs, cs = [], []
    for t in range(T):
        a = torch.addmm(b, torch.cat([x[t], h], 1), W)
        i, f, o, g = a[:, :H].sigmoid(), a[:, H:2 * H].sigmoid(), a[:, 2 * H:3 * H].sigmoid(), a[:, 3 * H:].tanh()
        c = f * c + i * g
        h = o * c.tanh()
        if tok is not None:
            keep = (tok[t] != 0).to(x.dtype)[:, None]
            h, c = h * keep, c * keep
        hs.append(h)
        cs.append(c)
    return torch.stack(hs), torch.stack(cs)


def torch_generic_forward(enc, dec, P, p, batch, drop):
    F = torch.nn.functional
    H, L, R = p['rnnHiddenSize'], p['numLayers'], p['maxQuesCount']
    f64 = torch.float64
    emb = P['embed'] * torch.cat([torch.zeros(1, 1, dtype=f64), torch.ones(P['embed'].shape[0] - 1, 1, dtype=f64)])
    tm = lambda a: torch.from_numpy(a.reshape(-1, a.shape[2]).T.astype(np.int64))
    dr = (lambda x, k: x * torch.from_numpy(drop[k]) * (1.0 / (1.0 - p['dropout']))) if drop else (lambda x, k: x)
    qtok = tm(batch['ques_fwd'])
    N = qtok.shape[1]
    B = N // R

    def stack(prefix, x, tok):
        st = []
        for l in range(L):
            h, c = torch_lstm2(x, P['%s%d.W' % (prefix, l + 1)], P['%s%d.b' % (prefix, l + 1)], tok, H)
            st.append((h, c))
            x = h
        return st
    img = None
    if 'img_feat' in batch and batch['img_feat'].ndim == 2:
        img = torch.from_numpy(batch['img_feat'].astype(np.float64)).repeat_interleave(R, 0)
    d5 = (lambda x, k: x * torch.from_numpy(drop[k]) * 2.0) if drop else (lambda x, k: x)   # hard-coded Dropout(0.5)
    qs = None
    if enc.startswith('lf-ques'):
        qs = stack('ques', emb[qtok], qtok)
        parts = [qs[-1][0][-1]]
        if 'im' in enc:
            parts.append(img)
        if 'hist' in enc:
            htok = tm(batch['hist'])
            parts.append(stack('hist', emb[htok], htok)[-1][0][-1])
        enc_out = torch.tanh(F.linear(dr(torch.cat(parts, 1), 'fuse'), P['fuse.W'], P['fuse.b']))
    elif enc.startswith('hre'):
        htok = tm(batch['hist'])
        hs = stack('hist', emb[htok], htok)
        x = emb[qtok]
        if 'im' in enc:
            im = d5(img, 'img') if enc.startswith('hrea') else img
            imgE = F.linear(im, P['img_embed.W'], P['img_embed.b'])
            xi = imgE[None].expand(qtok.shape[0], N, imgE.shape[1]) * (qtok != 0).to(f64)[:, :, None]
            x = torch.cat([x, xi], 2)
        qs = stack('ques', x, qtok)
        q, h = qs[-1][0][-1], hs[-1][0][-1]
        if enc.startswith('hrea'):
            sq = F.linear(q, P['att_q.W'], P['att_q.b']).view(B, R)
            sh = F.linear(h, P['att_h.W'], P['att_h.b']).view(B, R)
            A = sq[:, :, None] + sh[:, None, :]
            A = A.masked_fill(torch.triu(torch.ones(R, R, dtype=torch.bool), 1)[None], float('-inf'))
            att = torch.bmm(torch.softmax(A, -1), h.view(B, R, H)).reshape(N, H)
            j = torch.cat([att, q], 1)
        else:
            j = torch.cat([q, h], 1)
        dh, _ = torch_lstm2(j.view(B, R, 2 * H).transpose(0, 1), P['dialog.W'], P['dialog.b'], None, H)
        enc_out = dh.transpose(0, 1).reshape(N, H)
    elif enc in ('mn-ques-hist', 'mn-ques-im-hist', 'mn-att-ques-im-hist', 'lf-att-ques-im-hist'):
        htok = tm(batch['hist'])

        def two(prefix, tok, key):
            h1, _ = torch_lstm2(d5(emb[tok], key), P[prefix + '1.W'], P[prefix + '1.b'], tok, H)
            h2, _ = torch_lstm2(h1, P[prefix + '2.W'], P[prefix + '2.b'], tok, H)
            return h2[-1]
        h3, q3 = two('hist', htok, 'h_emb'), two('ques', qtok, 'q_emb')
        if enc == 'lf-att-ques-im-hist':
            u = torch.tanh(F.linear(torch.cat([q3, h3], 1), P['qh.W'], P['qh.b']))
        else:
            query = q3
            if enc == 'mn-ques-im-hist':
                query = torch.tanh(F.linear(torch.cat([q3, img], 1), P['qi.W'], P['qi.b']))
            sc = torch.bmm(query.view(B, R, H), h3.view(B, R, H).transpose(1, 2))
            sc = sc.masked_fill(torch.triu(torch.ones(R, R, dtype=torch.bool), 1)[None], -9999999.0)
            hatt = torch.bmm(torch.softmax(sc, -1), h3.view(B, R, H)).reshape(N, H)
            hattTr = torch.tanh(F.linear(d5(hatt, 'hatt'), P['mn1.W'], P['mn1.b']))
            u = torch.tanh(F.linear(hattTr + query, P['mn2.W'], P['mn2.b']))
        if 'att' in enc:
            S2 = p['imgSpatialSize'] ** 2
            im = torch.from_numpy(batch['img_feat'].astype(np.float64)).reshape(B, S2, -1)
            im = im[:, None].expand(B, R, S2, im.shape[-1]).reshape(N, S2, -1)
            img_tr = d5(torch.tanh(F.linear(im, P['img_proj.W'], P['img_proj.b'])), 'img_tr')
            ic = F.linear(img_tr, P['img_common.W'], P['img_common.b'])
            qc = F.linear(u, P['ques_common.W'], P['ques_common.b'])
            iqc = d5(torch.tanh(ic + qc[:, None, :]), 'iqc')
            patt = torch.softmax(F.linear(iqc, P['att.W'], P['att.b']).squeeze(-1), 1)
            u = torch.bmm(patt[:, None, :], img_tr).squeeze(1) + u
            enc_out = torch.tanh(F.linear(d5(u, 'u'), P['out.W'], P['out.b']))
        else:
            enc_out = u
    else:
        raise ValueError(enc)
    if dec == 'disc':
        opt = batch['options']
        O = opt.shape[1]
        otok = torch.from_numpy(opt.reshape(N * O, -1).T.astype(np.int64))
        oh, _ = torch_lstm2(emb[otok], P['opt.W'], P['opt.b'], None, H)
        scores = torch.bmm(oh[-1].view(N, O, H), enc_out[:, :, None]).squeeze(-1)
        return F.cross_entropy(scores, torch.from_numpy(batch['answer_ind'].astype(np.int64)) - 1)
    ain, aout = tm(batch['answer_in']), tm(batch['answer_out'])
    x = emb[ain]
    for l in range(L):
        h0 = c0 = None
        if qs is not None:
            h0, c0 = qs[l][0][-1], qs[l][1][-1]
        if l == L - 1:
            h0 = enc_out
        x, _ = torch_lstm2(x, P['dec%d.W' % (l + 1)], P['dec%d.b' % (l + 1)], ain, H, h0, c0)
    logp = F.log_softmax(F.linear(x, P['vocab.W'], P['vocab.b']), -1)
    keep = (ain != 0)
    tgt = torch.where(keep, aout - 1, torch.zeros_like(aout))
    nll = -logp.gather(2, tgt[:, :, None]).squeeze(-1)
    return (nll * keep.to(f64)).sum()


def fuse_masks(p, batch, enc, rng):
    from visdial_amd.dataloader import dropout_mask_shapes
    shp = dropout_mask_shapes(p, batch)
    if not shp:
        return None
    return {k: (rng.rand(*s) > (p['dropout'] if k == 'fuse' else 0.5)).astype(np.float64) for k, s in shp.items()}


@pytest.mark.parametrize("enc,dec", [('lf-ques', 'gen'), ('lf-ques-im-hist', 'gen'), ('hre-ques-im-hist', 'disc'),
                                     ('hre-ques-im-hist', 'gen'), ('lf-ques', 'disc'), ('lf-ques-im', 'gen'),
                                     ('lf-ques-hist', 'disc'), ('hre-ques-hist', 'gen'), ('hrea-ques-im-hist', 'disc'),
                                     ('hrea-ques-im-hist', 'gen'), ('mn-ques-hist', 'disc'), ('mn-ques-im-hist', 'gen'),
                                     ('mn-att-ques-im-hist', 'gen'), ('lf-att-ques-im-hist', 'disc'),
                                     ('lf-att-ques-im-hist', 'gen')])
@pytest.mark.parametrize("use_drop", [False, True])
def test_oracle_widening_matches_torch_autograd(enc, dec, use_drop):
    from visdial_amd.opts import derive
    p = derive(small_params(encoder=enc, decoder=dec, imgNorm=1, dropout=0.3, maxQuesCount=10 if enc.startswith('hrea') else 4))
    dl = SyntheticDataloader(p, seed=9)
    batch = dl.getTrainBatch(p)
    P = vo.init_params(enc, dec, p, seed=4)
    drop = fuse_masks(p, batch, enc, np.random.RandomState(2)) if use_drop else None
    r = vo.forward_backward(enc, dec, P, p, batch, drop)
    Pt = {k: torch.from_numpy(v.copy()).requires_grad_(True) for k, v in P.items()}
    loss = torch_generic_forward(enc, dec, Pt, p, batch, drop)
    loss.backward()
    assert abs(loss.item() - r['loss']) < 1e-9 * max(1.0, abs(r['loss']))
    for k in P:
        g_t = Pt[k].grad.numpy() if Pt[k].grad is not None else np.zeros_like(P[k])
        g_o = r['grads'][k]
        if k == 'embed':
            g_t, g_o = g_t[1:], g_o[1:]
        np.testing.assert_allclose(g_o, g_t, rtol=1e-7, atol=1e-9, err_msg=k)



@pytest.mark.parametrize("masked", [False, True])
def test_oracle_lstm_matches_torch_nn_lstm(masked):
    """The oracle's nn.SeqLSTM restatement against the THIRD-PARTY torch.nn.LSTM module (PyTorch's own LSTM kernel,
    not a loop written for this repo): weights permuted from SeqLSTM's [Wx ; Wh] x (i, f, o, g) layout to PyTorch's
    weight_ih / weight_hh with gate order (i, f, g, o).  Forward states and every gradient must agree.  maskZero is
    checked on right-aligned inputs (leading pads), where it is equivalent to starting the recurrence at the first
    real token with a zero state -- emulated for nn.LSTM with per-row packed sequences."""
    rng = np.random.RandomState(4)
    T, N, D, H = 7, 5, 6, 8
    x = rng.randn(T, N, D)
    W = rng.randn(D + H, 4 * H) / np.sqrt(D + H)
    b = rng.randn(4 * H) * 0.1
    tok = None
    lens = np.full(N, T)
    if masked:
        lens = rng.randint(1, T + 1, size=N)
        lens[0] = T
        tok = np.zeros((T, N), np.int64)
        for n in range(N):
            tok[T - lens[n]:, n] = 1
            x[:T - lens[n], n] = 0.0                                  # pad steps feed the zero vector
    dh_seq = rng.randn(T, N, H)
    if masked:
        dh_seq = dh_seq * (tok != 0)[:, :, None]                      # no gradient arrives at pad steps
    h, c, g = vo.lstm_forward(x, W, b, tok)
    dx, dW, db, _, _ = vo.lstm_backward(x, W, g, h, c, dh_seq=dh_seq)
    perm = np.concatenate([np.arange(0, H), np.arange(H, 2 * H), np.arange(3 * H, 4 * H), np.arange(2 * H, 3 * H)])  # i,f,o,g -> i,f,g,o
    lstm = torch.nn.LSTM(D, H).double()
    with torch.no_grad():
        lstm.weight_ih_l0.copy_(torch.from_numpy(W[:D][:, perm].T.copy()))
        lstm.weight_hh_l0.copy_(torch.from_numpy(W[D:][:, perm].T.copy()))
        lstm.bias_ih_l0.copy_(torch.from_numpy(b[perm].copy()))
        lstm.bias_hh_l0.zero_()
    xt = torch.from_numpy(x.copy()).requires_grad_(True)
    outs = []
    for n in range(N):                                                # right-aligned row = a shorter sequence
        o, _ = lstm(xt[T - lens[n]:, n:n + 1])
        outs.append(torch.cat([torch.zeros(T - lens[n], 1, H, dtype=torch.float64), o], 0))
    ht = torch.cat(outs, 1)
    np.testing.assert_allclose(ht.detach().numpy(), h, rtol=1e-10, atol=1e-12)
    (ht * torch.from_numpy(dh_seq)).sum().backward()
    np.testing.assert_allclose(xt.grad.numpy(), dx, rtol=1e-9, atol=1e-11)
    np.testing.assert_allclose(lstm.weight_ih_l0.grad.numpy().T[:, np.argsort(perm)], dW[:D], rtol=1e-9, atol=1e-11)
    np.testing.assert_allclose(lstm.weight_hh_l0.grad.numpy().T[:, np.argsort(perm)], dW[D:], rtol=1e-9, atol=1e-11)
    np.testing.assert_allclose(lstm.bias_ih_l0.grad.numpy()[np.argsort(perm)], db, rtol=1e-9, atol=1e-11)
