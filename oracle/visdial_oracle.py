"""CPU ORACLE -- TEST INFRASTRUCTURE ONLY.  Never imported by the product path (visdial_amd/).

PARITY UNPINNED: the reference (batra-mlp-lab/visdial) ships no tests, golden vectors or
known-answer fixtures, its arithmetic lives in un-vendored Torch7 rocks (torch, nn, nngraph,
Element-Research rnn @ git HEAD -- README.md:42-75) and no Lua/Torch7 runtime exists in the build
container, so this restatement cannot be checked against the reference executable.  It is pinned
instead by (a) an independent torch-autograd restatement of the same forward graphs
(tests/test_oracle.py) and (b) central finite differences in fp64.

Plain numpy restatement of the Visual-Dialog training step.  Each function cites the reference
file:line it follows (paths under /root/reference); semantics of the un-vendored nn/rnn modules
follow SURVEY.md Appendix A.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import numpy as np


# ----------------------------------------------------------------------------- primitives
def sigmoid(x):
    return 1.0 / (1.0 + np.exp(-x))


def lookup(emb, tok):
    """nn.LookupTableMaskZero (encoders/mn-att-ques-im-hist.lua:21): token 0 -> zero vector."""
    out = emb[tok]
    out = out * (tok != 0)[..., None]
    return out


def lookup_backward(demb, tok, dx):
    """accGradParameters of the shared table: dense gradWeight accumulate (pads hit row 0)."""
    np.add.at(demb, tok.reshape(-1), dx.reshape(-1, dx.shape[-1]))


def dropout(x, mask, p):
    """nn.Dropout in training mode with an explicit keep-mask (mask None = evaluate())."""
    if mask is None:
        return x
    return x * mask * (1.0 / (1.0 - p))


def linear(x, W, b):
    """nn.Linear: y = x W^T + b, weight [out x in]."""
    return x @ W.T + b


def linear_backward(x, W, dy):
    return dy @ W, dy.T @ x, dy.sum(0)


# ----------------------------------------------------------------------------- nn.SeqLSTM
def lstm_forward(x, W, b, tok_mask=None, h0=None, c0=None):
    """nn.SeqLSTM forward (Element-Research rnn; reference call sites mn-att:27-45,
    decoders/disc.lua:4-15).  x [T,N,D]; W [(D+H),4H] = [Wx;Wh]; gate order i,f,o,g.
    tok_mask [T,N] int: maskZero() -- rows whose token is 0 get h=c=gates=0 at that step."""
    T, N, D = x.shape
    H = W.shape[1] // 4
    Wx, Wh = W[:D], W[D:]
    dt = x.dtype
    h_all = np.zeros((T, N, H), dt)
    c_all = np.zeros((T, N, H), dt)
    gates = np.zeros((T, N, 4 * H), dt)
    hp = np.zeros((N, H), dt) if h0 is None else h0
    cp = np.zeros((N, H), dt) if c0 is None else c0
    for t in range(T):
        a = b + x[t] @ Wx + hp @ Wh
        i, f, o = sigmoid(a[:, :H]), sigmoid(a[:, H:2 * H]), sigmoid(a[:, 2 * H:3 * H])
        g = np.tanh(a[:, 3 * H:])
        c = f * cp + i * g
        h = o * np.tanh(c)
        gt = np.concatenate([i, f, o, g], 1)
        if tok_mask is not None:
            keep = (tok_mask[t] != 0)[:, None].astype(dt)
            h, c, gt = h * keep, c * keep, gt * keep
        h_all[t], c_all[t], gates[t] = h, c, gt
        hp, cp = h, c
    return h_all, c_all, gates


def lstm_backward(x, W, gates, h_all, c_all, dh_seq=None, dh_last=None, dc_last=None, h0=None, c0=None,
                  return_da=False):
    """nn.SeqLSTM backward.  Returns dx [T,N,D], dW, db, dh0, dc0."""
    T, N, D = x.shape
    H = W.shape[1] // 4
    Wx, Wh = W[:D], W[D:]
    dt = x.dtype
    dW = np.zeros_like(W)
    db = np.zeros(4 * H, dt)
    dx = np.zeros_like(x)
    dh_next = np.zeros((N, H), dt)
    dc = np.zeros((N, H), dt) if dc_last is None else dc_last.copy()
    da_all = np.zeros_like(gates)
    for t in range(T - 1, -1, -1):
        i, f, o, g = (gates[t][:, k * H:(k + 1) * H] for k in range(4))
        dh = dh_next.copy()
        if dh_seq is not None:
            dh = dh + dh_seq[t]
        if dh_last is not None and t == T - 1:
            dh = dh + dh_last
        tc = np.tanh(c_all[t])
        cprev = c_all[t - 1] if t > 0 else (np.zeros((N, H), dt) if c0 is None else c0)
        hprev = h_all[t - 1] if t > 0 else (np.zeros((N, H), dt) if h0 is None else h0)
        dc = dc + dh * o * (1 - tc * tc)
        da = np.concatenate([dc * g * i * (1 - i), dc * cprev * f * (1 - f), dh * tc * o * (1 - o),
                             dc * i * (1 - g * g)], 1)
        da_all[t] = da
        dx[t] = da @ Wx.T
        dW[:D] += x[t].T @ da
        dW[D:] += hprev.T @ da
        db += da.sum(0)
        dh_next = da @ Wh.T
        dc = dc * f
    if return_da:
        return dx, dW, db, dh_next, dc, da_all
    return dx, dW, db, dh_next, dc


# ----------------------------------------------------------------------------- memory attention
def mn_attention_forward(q, h, mask):
    """mn-att:48-62 + model_utils/MaskSoftMax.lua:5-21.  q,h [B,R,H]; mask [B,R,R] (1 = hidden)."""
    s = np.einsum('bik,bjk->bij', q, h)
    s = np.where(mask != 0, np.asarray(-9999999.0, s.dtype), s)
    s = s - s.max(-1, keepdims=True)
    e = np.exp(s)
    p = e / e.sum(-1, keepdims=True)
    return p, np.einsum('bij,bjk->bik', p, h)


def mn_attention_backward(q, h, p, dhatt):
    dp = np.einsum('bik,bjk->bij', dhatt, h)
    ds = p * (dp - (p * dp).sum(-1, keepdims=True))      # MaskSoftMax.lua:23-41 (THNN softmax bwd)
    dq = np.einsum('bij,bjk->bik', ds, h)
    dh = np.einsum('bij,bik->bjk', p, dhatt) + np.einsum('bij,bik->bjk', ds, q)
    return dq, dh


# ----------------------------------------------------------------------------- head / metrics
def cross_entropy(scores, gt):
    """nn.CrossEntropyCriterion (model.lua:37-38): mean over rows; gt 0-based here."""
    m = scores.max(1, keepdims=True)
    lse = m[:, 0] + np.log(np.exp(scores - m).sum(1))
    n = scores.shape[0]
    loss_rows = lse - scores[np.arange(n), gt]
    prob = np.exp(scores - lse[:, None])
    dscores = prob.copy()
    dscores[np.arange(n), gt] -= 1.0
    return loss_rows.mean(), dscores / n, loss_rows


def compute_ranks(scores, gt=None):
    """utils.lua:106-128: 1-based position in the descending sort (ties: lower index first)."""
    n, o = scores.shape
    order = np.argsort(-scores, axis=1, kind='stable')
    ranks = np.empty((n, o), np.int64)
    ranks[np.arange(n)[:, None], order] = np.arange(1, o + 1)[None, :]
    if gt is not None:
        return ranks[np.arange(n), gt]
    return ranks


def process_ranks(ranks, num_options=100):
    """utils.lua:131-160 (divides by the UNfiltered count; quirk D-12)."""
    ranks = np.asarray(ranks, np.float64).reshape(-1)
    num_ques = ranks.size
    ranks = ranks[ranks > 0]
    ranks = ranks[ranks <= num_options + 1]
    return {
        'numQues': num_ques,
        'r@1': float((ranks <= 1).sum()) / num_ques,
        'r@5': float((ranks <= 5).sum()) / num_ques,
        'r@10': float((ranks <= 10).sum()) / num_ques,
        'medianR': float(np.sort(ranks)[(ranks.size - 1) // 2]) if ranks.size else 0.0,  # torch.median = lower
        'meanR': float(ranks.mean()) if ranks.size else 0.0,
        'meanRR': float((1.0 / ranks).mean()) if ranks.size else 0.0,
    }


def clamp_adam(w, g, state, lr, clip=5.0, beta1=0.9, beta2=0.999, eps=1e-8):
    """model.lua:96-99 + model_utils/optim_updates.lua:62-91 (eps added to the uncorrected sqrt(v))."""
    g = np.clip(g, -clip, clip)
    if 'm' not in state:
        state['t'] = 0
        state['m'] = np.zeros_like(w)
        state['v'] = np.zeros_like(w)
    state['m'] = beta1 * state['m'] + (1 - beta1) * g
    state['v'] = beta2 * state['v'] + (1 - beta2) * g * g
    state['t'] += 1
    t = state['t']
    step = lr * np.sqrt(1 - beta2 ** t) / (1 - beta1 ** t)
    return w - (step * state['m'] / (np.sqrt(state['v']) + eps)).astype(w.dtype), g


# ----------------------------------------------------------------------------- parameters
def hop_suffixes(p, encoder=None):
    """attention hops: opts.lua:26 `numAttentionLayers` for mn-att (mn-att-ques-im-hist.lua:71); lf-att hard-codes ONE hop
    (lf-att-ques-im-hist.lua:49, `local num_attention_layer = 1`) whatever the option says"""
    L = int(p.get('numAttentionLayers', 1) or 1)
    if (encoder or p.get('encoder')) == 'lf-att-ques-im-hist':
        L = 1
    return [''] + [str(i) for i in range(2, L + 1)]


def param_spec(encoder, decoder, p):
    """Flat parameter layout (name, shape) shared by construction with visdial_amd/params.py.
    Tensors follow the reference modules: LSTM [(D+H) x 4H] + [4H], Linear [out x in] + [out],
    shared embedding [(V+1) x E] (SURVEY.md App. B parameter list)."""
    V, E, H = p['vocabSize'], p['embedSize'], p['rnnHiddenSize']
    spec = [('embed', (V + 1, E), 'embed')]

    def lstm(name, D):
        spec.append((name + '.W', (D + H, 4 * H), 'lstm_w'))
        spec.append((name + '.b', (4 * H,), 'lstm_b'))

    def lin(name, i, o):
        spec.append((name + '.W', (o, i), 'lin_w'))
        spec.append((name + '.b', (o,), 'lin_b'))

    def san():
        """mn-att:68-106: one img_common / ques_common / att triple PER attention hop (opts.lua:26), hop i > 1 named
        <name><i>"""
        C, K = p['imgFeatureSize'], p.get('commonEmbeddingSize', 512)
        lin('img_proj', C, H)
        for sfx in hop_suffixes(p, encoder):
            lin('img_common' + sfx, H, K); lin('ques_common' + sfx, H, K); lin('att' + sfx, K, 1)
        lin('out', H, H)

    if encoder == 'mn-att-ques-im-hist':
        lstm('hist1', E); lstm('hist2', H); lstm('ques1', E); lstm('ques2', H)
        lin('mn1', H, H); lin('mn2', H, H)
        san()
    elif encoder in ('mn-ques-hist', 'mn-ques-im-hist'):
        lstm('hist1', E); lstm('hist2', H); lstm('ques1', E); lstm('ques2', H)
        if encoder == 'mn-ques-im-hist':
            lin('qi', p['imgFeatureSize'] + H, H)
        lin('mn1', H, H); lin('mn2', H, H)
    elif encoder == 'lf-att-ques-im-hist':
        lstm('hist1', E); lstm('hist2', H); lstm('ques1', E); lstm('ques2', H)
        lin('qh', 2 * H, H)
        san()
    elif encoder in ('lf-ques', 'lf-ques-im', 'lf-ques-hist', 'lf-ques-im-hist'):
        for l in range(p['numLayers']):
            lstm('ques%d' % (l + 1), E if l == 0 else H)
        if 'hist' in encoder:
            for l in range(p['numLayers']):
                lstm('hist%d' % (l + 1), E if l == 0 else H)
        lin('fuse', H * (2 if 'hist' in encoder else 1) + (p['imgFeatureSize'] if 'im' in encoder else 0), H)
    elif encoder in ('hre-ques-hist', 'hre-ques-im-hist', 'hrea-ques-im-hist'):
        DI = p['imgEmbedSize'] if 'im' in encoder else 0
        for l in range(p['numLayers']):
            lstm('hist%d' % (l + 1), E if l == 0 else H)
        if DI:
            lin('img_embed', p['imgFeatureSize'], DI)
        for l in range(p['numLayers']):
            lstm('ques%d' % (l + 1), (E + DI) if l == 0 else H)
        if encoder.startswith('hrea'):
            lin('att_q', H, 1); lin('att_h', H, 1)
        lstm('dialog', 2 * H)
    else:
        raise ValueError('oracle: encoder %s not restated yet' % encoder)
    if decoder == 'disc':
        lstm('opt', E)
    elif decoder == 'gen':
        for l in range(p['numLayers']):
            lstm('dec%d' % (l + 1), E if l == 0 else H)
        lin('vocab', H, V)
    else:
        raise ValueError(decoder)
    return spec


def init_params(encoder, decoder, p, seed=1234, dtype=np.float64):
    """Library-default init (weight-init.lua is a no-op):
    SeqLSTM weight ~ N(0, 1/sqrt(D+H)), bias 0 with forget gate 1; Linear U(+-1/sqrt(in));
    LookupTable N(0,1) with the pad row zero."""
    rng = np.random.RandomState(seed)
    H = p['rnnHiddenSize']
    out = {}
    for name, shape, kind in param_spec(encoder, decoder, p):
        if kind == 'embed':
            w = rng.randn(*shape)
            w[0] = 0
        elif kind == 'lstm_w':
            w = rng.randn(*shape) * (1.0 / np.sqrt(shape[0]))
        elif kind == 'lstm_b':
            w = np.zeros(shape)
            w[H:2 * H] = 1.0
        elif kind == 'lin_w':
            s = 1.0 / np.sqrt(shape[1])
            w = rng.uniform(-s, s, shape)
        elif kind == 'lin_b':
            s = 1.0 / np.sqrt(out[name[:-2] + '.W'].shape[1])
            w = rng.uniform(-s, s, shape)
        else:
            raise AssertionError(kind)
        out[name] = w.astype(dtype)
    return out


def flatten(params, spec):
    return np.concatenate([params[e[0]].reshape(-1) for e in spec])


def unflatten(vec, spec):
    out, o = {}, 0
    for e in spec:
        n, s = e[0], e[1]
        k = int(np.prod(s))
        out[n] = vec[o:o + k].reshape(s)
        o += k
    return out


# ----------------------------------------------------------------------------- model step
def causal_mask(B, R):
    """model.lua:280-294: mask[i][j] = 0 iff j <= i, tiled over the batch.  -> [B,R,R] uint8"""
    m = (np.arange(R)[None, :] > np.arange(R)[:, None]).astype(np.uint8)
    return np.broadcast_to(m, (B, R, R)).copy()


def _two_layer_lstm_fwd(P, n1, n2, x, tok):
    h1, c1, g1 = lstm_forward(x, P[n1 + '.W'], P[n1 + '.b'], tok)
    h2, c2, g2 = lstm_forward(h1, P[n2 + '.W'], P[n2 + '.b'], tok)
    return dict(x=x, h1=h1, c1=c1, g1=g1, h2=h2, c2=c2, g2=g2)


def _two_layer_lstm_bwd(P, G, n1, n2, S, dlast):
    dx2, dW2, db2, _, _ = lstm_backward(S['h1'], P[n2 + '.W'], S['g2'], S['h2'], S['c2'], dh_last=dlast)
    dx1, dW1, db1, _, _ = lstm_backward(S['x'], P[n1 + '.W'], S['g1'], S['h1'], S['c1'], dh_seq=dx2)
    G[n2 + '.W'] += dW2; G[n2 + '.b'] += db2
    G[n1 + '.W'] += dW1; G[n1 + '.b'] += db1
    return dx1


def _mn_block_fwd(P, query, h3, B, R, d):
    """memory attention + the two Linear+Tanh of mn-att:48-65 / mn-ques-hist.lua:43-58.
    query [N,H] is q3 (mn-ques-hist, mn-att) or qi_proj (mn-ques-im-hist)."""
    N, H = query.shape
    prob, hatt = mn_attention_forward(query.reshape(B, R, H), h3.reshape(B, R, H), causal_mask(B, R))
    hatt_d = dropout(hatt.reshape(N, H), d('hatt'), 0.5)
    hattTr = np.tanh(linear(hatt_d, P['mn1.W'], P['mn1.b']))
    s2 = hattTr + query
    qh2 = np.tanh(linear(s2, P['mn2.W'], P['mn2.b']))
    return qh2, dict(query=query, h3=h3, prob=prob, hatt_d=hatt_d, hattTr=hattTr, s2=s2, qh2=qh2, B=B, R=R)


def _mn_block_bwd(P, G, st, dqh2, dback):
    N, H = st['query'].shape
    B, R = st['B'], st['R']
    ds2p = dqh2 * (1 - st['qh2'] ** 2)
    ds2, dW, db = linear_backward(st['s2'], P['mn2.W'], ds2p)
    G['mn2.W'] += dW; G['mn2.b'] += db
    dhTr_p = ds2 * (1 - st['hattTr'] ** 2)
    dhatt_d, dW, db = linear_backward(st['hatt_d'], P['mn1.W'], dhTr_p)
    G['mn1.W'] += dW; G['mn1.b'] += db
    dhatt = dback(dhatt_d, 'hatt')
    dqv, dhv = mn_attention_backward(st['query'].reshape(B, R, H), st['h3'].reshape(B, R, H), st['prob'],
                                     dhatt.reshape(B, R, H))
    return ds2 + dqv.reshape(N, H), dhv.reshape(N, H)


def _san_block_fwd(P, p, batch, u0, R, d):
    """SAN image attention, numAttentionLayers hops, + output layer (mn-att:68-106 / lf-att-ques-im-hist.lua:45-86).
    All hops attend over the same img_tr; every hop has its own Linears and its own Dropout (mask 'iqc', 'iqc2', ..)."""
    N, H = u0.shape
    B = N // R
    S2 = p['imgSpatialSize'] ** 2
    img = batch['img_feat'].reshape(B * S2, p['imgFeatureSize'])
    pre = np.tanh(linear(img, P['img_proj.W'], P['img_proj.b']))            # per image
    pre_r = np.repeat(pre.reshape(B, 1, S2, H), R, 1).reshape(N, S2, H)     # model.lua:262-265 repeat
    img_tr = dropout(pre_r, d('img_tr'), 0.5)
    hops = []
    u = u0
    for sfx in hop_suffixes(p):
        img_common = linear(img_tr.reshape(N * S2, H), P['img_common%s.W' % sfx], P['img_common%s.b' % sfx]).reshape(N, S2, -1)
        qc = linear(u, P['ques_common%s.W' % sfx], P['ques_common%s.b' % sfx])
        t_iqc = np.tanh(img_common + qc[:, None, :])
        iqc = dropout(t_iqc, d('iqc' + sfx), 0.5)
        score = (iqc @ P['att%s.W' % sfx][0]) + P['att%s.b' % sfx][0]
        score = score - score.max(1, keepdims=True)
        e = np.exp(score)
        patt = e / e.sum(1, keepdims=True)
        att = np.einsum('ns,nsh->nh', patt, img_tr)
        hops.append(dict(u_in=u, t_iqc=t_iqc, iqc=iqc, patt=patt, sfx=sfx))
        u = att + u
    u1_d = dropout(u, d('u'), 0.5)
    out = np.tanh(linear(u1_d, P['out.W'], P['out.b']))
    return out, dict(u0=u0, img=img, pre=pre, img_tr=img_tr, hops=hops, u1_d=u1_d, out=out, R=R)


def _san_block_bwd(P, G, p, st, denc, dback):
    N, H = st['u0'].shape
    R = st['R']
    B = N // R
    S2 = p['imgSpatialSize'] ** 2
    dpre_o = denc * (1 - st['out'] ** 2)
    du1d, dW, db = linear_backward(st['u1_d'], P['out.W'], dpre_o)
    G['out.W'] += dW; G['out.b'] += db
    du = dback(du1d, 'u')
    img_tr = st['img_tr']
    dimg_tr = np.zeros_like(img_tr)
    for hp in reversed(st['hops']):
        sfx, patt, iqc = hp['sfx'], hp['patt'], hp['iqc']
        datt = du
        dp = np.einsum('nh,nsh->ns', datt, img_tr)
        dimg_tr = dimg_tr + patt[:, :, None] * datt[:, None, :]
        dscore = patt * (dp - (patt * dp).sum(1, keepdims=True))
        G['att%s.W' % sfx][0] += np.einsum('ns,nsk->k', dscore, iqc)
        G['att%s.b' % sfx][0] += dscore.sum()
        diqc = dscore[:, :, None] * P['att%s.W' % sfx][0][None, None, :]
        dz = dback(diqc, 'iqc' + sfx) * (1 - hp['t_iqc'] ** 2)
        dqc = dz.sum(1)
        dimg_tr2, dW, db = linear_backward(img_tr.reshape(N * S2, H), P['img_common%s.W' % sfx], dz.reshape(N * S2, -1))
        G['img_common%s.W' % sfx] += dW; G['img_common%s.b' % sfx] += db
        dimg_tr = dimg_tr + dimg_tr2.reshape(N, S2, H)
        dq, dW, db = linear_backward(hp['u_in'], P['ques_common%s.W' % sfx], dqc)
        G['ques_common%s.W' % sfx] += dW; G['ques_common%s.b' % sfx] += db
        du = du + dq                                                     # residual CAddTable + the ques_common path
    dpre = dback(dimg_tr, 'img_tr').reshape(B, R, S2, H).sum(1).reshape(B * S2, H)
    dpre_a = dpre * (1 - st['pre'] ** 2)
    _, dW, db = linear_backward(st['img'], P['img_proj.W'], dpre_a)
    G['img_proj.W'] += dW; G['img_proj.b'] += db
    return du


GRAPH_ENCODERS = ('mn-ques-hist', 'mn-ques-im-hist', 'mn-att-ques-im-hist', 'lf-att-ques-im-hist')


def graph_encoder_forward(encoder, P, p, batch, drop):
    """the four nngraph encoders: two text branches (embed -> Dropout(0.5) -> 2 x SeqLSTM(maskZero), hard-coded
    in every file, e.g. mn-att:24-45) followed by memory attention and/or SAN image attention."""
    B, R, Tq = batch['ques_fwd'].shape
    N = B * R
    H = p['rnnHiddenSize']
    d = (lambda k: None) if drop is None else (lambda k: drop[k])
    qtok, htok = _tm(batch['ques_fwd']), _tm(batch['hist'])
    st = dict(qtok=qtok, htok=htok)
    st['hs'] = _two_layer_lstm_fwd(P, 'hist1', 'hist2', dropout(lookup(P['embed'], htok), d('h_emb'), 0.5), htok)
    st['qs'] = _two_layer_lstm_fwd(P, 'ques1', 'ques2', dropout(lookup(P['embed'], qtok), d('q_emb'), 0.5), qtok)
    h3, q3 = st['hs']['h2'][-1], st['qs']['h2'][-1]
    if encoder == 'lf-att-ques-im-hist':                                   # lf-att:43
        st['cat'] = np.concatenate([q3, h3], 1)
        st['qh'] = np.tanh(linear(st['cat'], P['qh.W'], P['qh.b']))
        u = st['qh']
    else:
        query = q3
        if encoder == 'mn-ques-im-hist':                                   # mn-ques-im-hist.lua:47-48
            st['qi'] = np.concatenate([q3, _img_rep(batch, R)], 1)
            st['qi_proj'] = np.tanh(linear(st['qi'], P['qi.W'], P['qi.b']))
            query = st['qi_proj']
        u, st['mn'] = _mn_block_fwd(P, query, h3, B, R, d)
    if 'att' in encoder:
        out, st['san'] = _san_block_fwd(P, dict(p, encoder=encoder), batch, u, R, d)
    else:
        out = u
    st['enc_out'] = out
    return out, st


def graph_encoder_backward(encoder, P, G, p, batch, drop, st, denc):
    H = p['rnnHiddenSize']
    d = (lambda k: None) if drop is None else (lambda k: drop[k])

    def dback(dy, key):
        m = d(key)
        return dy if m is None else dy * m * 2.0
    du = _san_block_bwd(P, G, p, st['san'], denc, dback) if 'att' in encoder else denc
    if encoder == 'lf-att-ques-im-hist':
        dcat, dW, db = linear_backward(st['cat'], P['qh.W'], du * (1 - st['qh'] ** 2))
        G['qh.W'] += dW; G['qh.b'] += db
        dq3, dh3 = dcat[:, :H], dcat[:, H:]
    else:
        dquery, dh3 = _mn_block_bwd(P, G, st['mn'], du, dback)
        if encoder == 'mn-ques-im-hist':
            dqi, dW, db = linear_backward(st['qi'], P['qi.W'], dquery * (1 - st['qi_proj'] ** 2))
            G['qi.W'] += dW; G['qi.b'] += db
            dq3 = dqi[:, :H]
        else:
            dq3 = dquery
    dqx = _two_layer_lstm_bwd(P, G, 'ques1', 'ques2', st['qs'], dq3)
    dhx = _two_layer_lstm_bwd(P, G, 'hist1', 'hist2', st['hs'], dh3)
    lookup_backward(G['embed'], st['qtok'], dback(dqx, 'q_emb'))
    lookup_backward(G['embed'], st['htok'], dback(dhx, 'h_emb'))


def mnatt_encoder_forward(P, p, batch, drop):
    return graph_encoder_forward('mn-att-ques-im-hist', P, p, batch, drop)


def mnatt_encoder_backward(P, G, p, batch, drop, st, denc):
    return graph_encoder_backward('mn-att-ques-im-hist', P, G, p, batch, drop, st, denc)


def disc_decoder_forward(P, p, options, enc_out):
    """decoders/disc.lua:3-32: options [N,O,T] (left-aligned, trailing zeros, NO maskZero)."""
    N, O, T = options.shape
    otok = options.reshape(N * O, T).T
    ox = lookup(P['embed'], otok)
    h, c, g = lstm_forward(ox, P['opt.W'], P['opt.b'], None)
    optH = h[-1].reshape(N, O, -1)
    scores = np.einsum('noh,nh->no', optH, enc_out)
    return scores, dict(otok=otok, ox=ox, h=h, c=c, g=g, optH=optH)


def disc_decoder_backward(P, G, st, enc_out, dscores):
    optH = st['optH']
    N, O, H = optH.shape
    doptH = dscores[:, :, None] * enc_out[:, None, :]
    denc = np.einsum('no,noh->nh', dscores, optH)
    dx, dW, db, _, _ = lstm_backward(st['ox'], P['opt.W'], st['g'], st['h'], st['c'], dh_last=doptH.reshape(N * O, H))
    G['opt.W'] += dW; G['opt.b'] += db
    lookup_backward(G['embed'], st['otok'], dx)
    return denc


# ----------------------------------------------------------------------------- lf / hre encoders
def _stack_fwd(P, names, x, tok, h0s=None, c0s=None):
    """numLayers x SeqLSTM(maskZero) (encoders/lf-ques.lua:17-24)."""
    states = []
    for i, nm in enumerate(names):
        h0 = None if h0s is None else h0s[i]
        c0 = None if c0s is None else c0s[i]
        h, c, g = lstm_forward(x, P[nm + '.W'], P[nm + '.b'], tok, h0, c0)
        states.append(dict(x=x, h=h, c=c, g=g, h0=h0, c0=c0))
        x = h
    return states


def _stack_bwd(P, G, names, states, dh_last_top=None, dh_seq_top=None, dh_last=None, dc_last=None):
    """dh_last / dc_last: optional per-layer lists (gradPrevOutput / userNextGradCell, gen.lua:49-51).
    Returns (dx of the bottom layer, [dh0 per layer], [dc0 per layer])."""
    L = len(names)
    dseq = dh_seq_top
    dh0s, dc0s = [None] * L, [None] * L
    for i in range(L - 1, -1, -1):
        st = states[i]
        dl = dh_last_top if i == L - 1 else None
        if dh_last is not None and dh_last[i] is not None:
            dl = dh_last[i] if dl is None else dl + dh_last[i]
        dcl = None if dc_last is None else dc_last[i]
        dx, dW, db, dh0, dc0 = lstm_backward(st['x'], P[names[i] + '.W'], st['g'], st['h'], st['c'], dh_seq=dseq,
                                             dh_last=dl, dc_last=dcl, h0=st['h0'], c0=st['c0'])
        G[names[i] + '.W'] += dW
        G[names[i] + '.b'] += db
        dh0s[i], dc0s[i] = dh0, dc0
        dseq = dx
    return dx, dh0s, dc0s


def _layer_names(prefix, p):
    return ['%s%d' % (prefix, l + 1) for l in range(p['numLayers'])]


def _tm(a):
    """[B,R,T] -> time-major [T, N] (model.lua:255-257)"""
    return a.reshape(-1, a.shape[2]).T


def _img_rep(batch, R):
    f = batch['img_feat']
    return np.repeat(f, R, axis=0)                      # model.lua:267-269


def encoder_forward(encoder, P, p, batch, drop):
    if encoder in GRAPH_ENCODERS:
        return graph_encoder_forward(encoder, P, p, batch, drop)
    d = (lambda k: None) if drop is None else (lambda k: drop[k])
    pd = p.get('dropout', 0.5)
    H = p['rnnHiddenSize']
    qtok = _tm(batch['ques_fwd'])
    st = dict(qtok=qtok)
    if encoder in ('lf-ques', 'lf-ques-im', 'lf-ques-hist', 'lf-ques-im-hist'):
        # encoders/lf-ques.lua:3-36, lf-ques-im.lua, lf-ques-hist.lua, lf-ques-im-hist.lua:3-62
        st['qs'] = _stack_fwd(P, _layer_names('ques', p), lookup(P['embed'], qtok), qtok)
        parts = [st['qs'][-1]['h'][-1]]
        if 'im' in encoder:
            parts.append(_img_rep(batch, p['maxQuesCount']))
        if 'hist' in encoder:
            htok = _tm(batch['hist'])
            st['htok'] = htok
            st['hs'] = _stack_fwd(P, _layer_names('hist', p), lookup(P['embed'], htok), htok)
            parts.append(st['hs'][-1]['h'][-1])
        st['cat'] = dropout(np.concatenate(parts, 1), d('fuse'), pd)
        st['out'] = np.tanh(linear(st['cat'], P['fuse.W'], P['fuse.b']))
    elif encoder in ('hre-ques-hist', 'hre-ques-im-hist', 'hrea-ques-im-hist'):
        # encoders/hre-ques-hist.lua, hre-ques-im-hist.lua:5-97, hrea-ques-im-hist.lua
        htok = _tm(batch['hist'])
        st['htok'] = htok
        R = p['maxQuesCount']
        N = qtok.shape[1]
        B = N // R
        st['hs'] = _stack_fwd(P, _layer_names('hist', p), lookup(P['embed'], htok), htok)
        x = lookup(P['embed'], qtok)
        if 'im' in encoder:
            st['img_in'] = _img_rep(batch, R)
            if encoder.startswith('hrea'):
                st['img_in'] = dropout(st['img_in'], d('img'), 0.5)                              # hrea:47
            imgE = linear(st['img_in'], P['img_embed.W'], P['img_embed.b'])                      # hre:43-48
            st['keep'] = (qtok != 0)[:, :, None].astype(imgE.dtype)
            x = np.concatenate([x, imgE[None, :, :] * st['keep']], 2)                           # MaskTime + JoinTable(-1)
        st['qs'] = _stack_fwd(P, _layer_names('ques', p), x, qtok)
        q, h = st['qs'][-1]['h'][-1], st['hs'][-1]['h'][-1]
        first = q
        if encoder.startswith('hrea'):                                                           # hrea:83-131
            sq = linear(q, P['att_q.W'], P['att_q.b']).reshape(B, R)
            sh = linear(h, P['att_h.W'], P['att_h.b']).reshape(B, R)
            A = sq[:, :, None] + sh[:, None, :]                                                  # Replicate + CAddTable
            A = np.where(np.triu(np.ones((R, R), bool), 1)[None], 0.0, A)                        # MaskFuture
            A = np.where(A == 0, -np.inf, A)                                                     # ReplaceZero(-inf)
            A = A - A.max(-1, keepdims=True)
            e = np.exp(A)
            Pm = e / e.sum(-1, keepdims=True)
            hv = h.reshape(B, R, H)
            att = np.einsum('bij,bjk->bik', Pm, hv).reshape(N, H)                                # CMul + Sum(3)
            st.update(P_att=Pm, q=q, h=h)
            j = np.concatenate([att, q], 1)                                                      # concat4: {att, ques}
        else:
            j = np.concatenate([q, h], 1)
        dx = j.reshape(B, R, 2 * H).transpose(1, 0, 2)
        st['ds'] = _stack_fwd(P, ['dialog'], np.ascontiguousarray(dx), None)
        st['out'] = st['ds'][0]['h'].transpose(1, 0, 2).reshape(N, H)
    else:
        raise ValueError('oracle: encoder %s not restated yet' % encoder)
    return st['out'], st


def encoder_backward(encoder, P, G, p, batch, drop, st, denc, dec_dh0=None, dec_dc0=None):
    """dec_dh0 / dec_dc0: per-layer gradients handed back by the gen decoder (gen.lua:45-60)."""
    if encoder in GRAPH_ENCODERS:
        return graph_encoder_backward(encoder, P, G, p, batch, drop, st, denc)
    d = (lambda k: None) if drop is None else (lambda k: drop[k])
    pd = p.get('dropout', 0.5)
    H = p['rnnHiddenSize']
    L = p['numLayers']
    dh_last = dc_last = None
    if dec_dh0 is not None:
        dh_last = [dec_dh0[i] if i != L - 1 else None for i in range(L)]     # gradPrevOutput, ii ~= top
        dc_last = list(dec_dc0)                                              # userNextGradCell
    if encoder in ('lf-ques', 'lf-ques-im', 'lf-ques-hist', 'lf-ques-im-hist'):
        dpre = denc * (1 - st['out'] ** 2)
        dcat, dW, db = linear_backward(st['cat'], P['fuse.W'], dpre)
        G['fuse.W'] += dW; G['fuse.b'] += db
        m = d('fuse')
        if m is not None:
            dcat = dcat * m * (1.0 / (1.0 - pd))
        dq = dcat[:, :H]
        dx, _, _ = _stack_bwd(P, G, _layer_names('ques', p), st['qs'], dh_last_top=dq, dh_last=dh_last, dc_last=dc_last)
        lookup_backward(G['embed'], st['qtok'], dx)
        if 'hist' in encoder:
            dh = dcat[:, -H:]
            dx, _, _ = _stack_bwd(P, G, _layer_names('hist', p), st['hs'], dh_last_top=dh)
            lookup_backward(G['embed'], st['htok'], dx)
    elif encoder in ('hre-ques-hist', 'hre-ques-im-hist', 'hrea-ques-im-hist'):
        R = p['maxQuesCount']
        N = denc.shape[0]
        B = N // R
        E = p['embedSize']
        g = np.ascontiguousarray(denc.reshape(B, R, H).transpose(1, 0, 2))
        dj, _, _ = _stack_bwd(P, G, ['dialog'], st['ds'], dh_seq_top=g)
        dj = dj.transpose(1, 0, 2).reshape(N, 2 * H)
        if encoder.startswith('hrea'):
            datt, dq = dj[:, :H].reshape(B, R, H), dj[:, H:].copy()
            Pm, hv = st['P_att'], st['h'].reshape(B, R, H)
            dP = np.einsum('bik,bjk->bij', datt, hv)
            dA = Pm * (dP - (Pm * dP).sum(-1, keepdims=True))
            dh = np.einsum('bij,bik->bjk', Pm, datt).reshape(N, H)
            dsq, dsh = dA.sum(2).reshape(N, 1), dA.sum(1).reshape(N, 1)
            dq_, dW, db = linear_backward(st['q'], P['att_q.W'], dsq)
            G['att_q.W'] += dW; G['att_q.b'] += db
            dh_, dW, db = linear_backward(st['h'], P['att_h.W'], dsh)
            G['att_h.W'] += dW; G['att_h.b'] += db
            dq, dh = dq + dq_, dh + dh_
        else:
            dq, dh = dj[:, :H], dj[:, H:]
        dx, _, _ = _stack_bwd(P, G, _layer_names('ques', p), st['qs'], dh_last_top=dq, dh_last=dh_last, dc_last=dc_last)
        lookup_backward(G['embed'], st['qtok'], dx[:, :, :E])
        if 'im' in encoder:
            dimgE = (dx[:, :, E:] * st['keep']).sum(0)                                           # MaskTime.lua:31-40
            _, dW, db = linear_backward(st['img_in'], P['img_embed.W'], dimgE)
            G['img_embed.W'] += dW; G['img_embed.b'] += db
        dx, _, _ = _stack_bwd(P, G, _layer_names('hist', p), st['hs'], dh_last_top=dh)
        lookup_backward(G['embed'], st['htok'], dx)


def gen_decoder_forward(P, p, batch, enc_out, enc_state):
    """decoders/gen.lua:3-42 + criterion of model.lua:32-36,306-314."""
    L = p['numLayers']
    ain, aout = _tm(batch['answer_in']), _tm(batch['answer_out'])
    N = ain.shape[1]
    h0s, c0s = [None] * L, [None] * L
    qs = enc_state.get('qs') if isinstance(enc_state.get('qs'), list) else None   # Sequential encoders expose rnnLayers
    if qs is not None:
        for i in range(L):
            h0s[i], c0s[i] = qs[i]['h'][-1], qs[i]['c'][-1]                        # gen.lua:32-35
        h0s[L - 1] = enc_out                                                       # gen.lua:38
    else:
        h0s[L - 1] = enc_out                                                       # gen.lua:40
        c0s[L - 1] = np.zeros_like(enc_out)
    ds = _stack_fwd(P, _layer_names('dec', p), lookup(P['embed'], ain), ain, h0s, c0s)
    h = ds[-1]['h']
    logits = h @ P['vocab.W'].T + P['vocab.b']
    m = logits.max(-1, keepdims=True)
    lse = m[..., 0] + np.log(np.exp(logits - m).sum(-1))
    keep = (ain != 0)
    tgt = np.where(keep, aout - 1, 0)
    T, _ = ain.shape
    picked = logits[np.arange(T)[:, None], np.arange(N)[None, :], tgt]
    loss = float(((lse - picked) * keep).sum())
    dlogits = np.exp(logits - lse[..., None])
    dlogits[np.arange(T)[:, None], np.arange(N)[None, :], tgt] -= 1.0
    dlogits = dlogits * keep[..., None]
    return loss, dict(ds=ds, h=h, dlogits=dlogits, ain=ain, shared=qs is not None)


def gen_decoder_backward(P, G, p, st):
    L = p['numLayers']
    h, dl = st['h'], st['dlogits']
    T, N, H = h.shape
    G['vocab.W'] += dl.reshape(T * N, -1).T @ h.reshape(T * N, H)
    G['vocab.b'] += dl.reshape(T * N, -1).sum(0)
    dh = dl @ P['vocab.W']
    dx, dh0s, dc0s = _stack_bwd(P, G, _layer_names('dec', p), st['ds'], dh_seq_top=dh)
    lookup_backward(G['embed'], st['ain'], dx)
    denc = dh0s[L - 1]                                                            # gen.lua:56 / :58
    if st['shared']:
        return denc, dh0s, dc0s
    return denc, None, None


def compute_lhood(words, logp):
    """utils.lua:86-102: sum over time of log p(word), pads (word == 0) contribute 0.  words [T,N], logp [T,N,V]."""
    T, N = words.shape
    keep = words != 0
    idx = np.where(keep, words - 1, 0)
    picked = logp[np.arange(T)[:, None], np.arange(N)[None, :], idx]
    return (picked * keep).sum(0)


def gen_retrieve_scores(P, p, batch, enc_out, enc_state):
    """Model:retrieveBatch, gen branch (model.lua:392-420): for every candidate option run the decoder from
    the encoder state (forwardConnect) and score it with its summed token log-likelihood.  -> [N, O]"""
    L = p['numLayers']
    oin, oout = batch['option_in'], batch['option_out']            # [B,R,O,T]
    B, R, O, T = oin.shape
    N = B * R
    oin = oin.reshape(N, O, T).transpose(1, 2, 0)                   # model.lua:394-399: [O, T, N]
    oout = oout.reshape(N, O, T).transpose(1, 2, 0)
    qs = enc_state.get('qs') if isinstance(enc_state.get('qs'), list) else None
    lhood = np.zeros((O, N), enc_out.dtype)
    for o in range(O):
        h0s, c0s = [None] * L, [None] * L
        if qs is not None:
            for i in range(L):
                h0s[i], c0s[i] = qs[i]['h'][-1], qs[i]['c'][-1]
            h0s[L - 1] = enc_out
        else:
            h0s[L - 1], c0s[L - 1] = enc_out, np.zeros_like(enc_out)
        ds = _stack_fwd(P, _layer_names('dec', p), lookup(P['embed'], oin[o]), oin[o], h0s, c0s)
        logits = ds[-1]['h'] @ P['vocab.W'].T + P['vocab.b']
        m = logits.max(-1, keepdims=True)
        logp = logits - (m + np.log(np.exp(logits - m).sum(-1, keepdims=True)))
        logp = logp * (oin[o] != 0)[:, :, None]                     # MaskZero(LogSoftMax): pad steps give zero rows
        lhood[o] = compute_lhood(oout[o], logp)
    return lhood.T


def generate_beam(encoder, P, p, batch, beam_size, beam_len, START, END):
    """Model:generateAnswers, beam-search branch (model.lua:466-573), for ONE dialog: returns, per round,
    (best token vector [beam_len], its score).  Candidate ties are broken by insertion order (stable sort)."""
    P = dict(P)
    P['embed'] = P['embed'].copy()
    P['embed'][0] = 0
    L = p['numLayers']
    names = _layer_names('dec', p)
    enc_out, st = encoder_forward(encoder, P, p, batch, None)
    qs = st.get('qs') if isinstance(st.get('qs'), list) else None
    R = batch['ques_fwd'].shape[1]
    H = p['rnnHiddenSize']
    out = []
    for it in range(R):
        hid = []
        for lv in range(L):
            if qs is not None:
                h = enc_out[it] if lv == L - 1 else qs[lv]['h'][-1][it]
                c = qs[lv]['c'][-1][it]
            else:
                h = enc_out[it] if lv == L - 1 else np.zeros(H)
                c = np.zeros(H)
            hid.append((np.tile(h, (beam_size, 1)), np.tile(c, (beam_size, 1))))
        beams = np.zeros((beam_len, beam_size), np.int64)
        beams[0] = START
        scores = np.zeros(beam_size)
        finish = []
        for step in range(1, beam_len):
            explore = 1 if step == 1 else beam_size
            tok = beams[step - 1:step]
            x = lookup(P['embed'], tok)
            newh = []
            for lv in range(L):
                h, c, _ = lstm_forward(x, P[names[lv] + '.W'], P[names[lv] + '.b'], tok, hid[lv][0], hid[lv][1])
                newh.append((h[0], c[0]))
                x = h
            logits = x[0] @ P['vocab.W'].T + P['vocab.b']
            m = logits.max(-1, keepdims=True)
            logp = logits - (m + np.log(np.exp(logits - m).sum(-1, keepdims=True)))
            # decoders/gen.lua:23-24: Sequencer(MaskZero(Linear)), Sequencer(MaskZero(LogSoftMax)) -- a hypothesis whose last token is 0
            # (a beam slot that was never filled: fewer than beamSize unfinished candidates so far, model.lua:560) has h = c = 0 from
            # maskZero AND an all-ZERO row of "log-probabilities": its children cost nothing.  torch.topk over that constant row is
            # implementation-defined in Torch7; here (and in every host of this repo) ties go to the lower index.
            logp[tok[0] == 0] = 0.0
            cands = []
            for w in range(explore):
                for cid in np.argsort(-logp[w], kind='stable')[:beam_size]:
                    cb = beams[:, w].copy()
                    cb[step] = cid + 1
                    sc = scores[w] + logp[w, cid]
                    if cid + 1 == END:
                        finish.append((sc, cb))
                    else:
                        cands.append((sc, cb, w))
            cands.sort(key=lambda a: -a[0])
            for i, (sc, cb, w) in enumerate(cands[:beam_size]):
                beams[:, i] = cb
                scores[i] = sc
                for lv in range(L):
                    hid[lv][0][i] = newh[lv][0][w]
                    hid[lv][1][i] = newh[lv][1][w]
            # NOTE hid rows are overwritten in candidate order exactly like model.lua:556-566 (in-place, so a
            # later candidate may copy from a slot an earlier one already replaced only via newh, which is a copy)
        finish.sort(key=lambda a: -a[0])
        out.append((finish[0][1], finish[0][0]) if finish else (beams[:, 0], scores[0]))
    return out


def retrieve(encoder, decoder, P, p, batch):
    """evaluate()-mode scores of the 100 candidates of every round: disc -> dot products, gen -> likelihoods."""
    P = dict(P)
    P['embed'] = P['embed'].copy()
    P['embed'][0] = 0
    enc_out, st = encoder_forward(encoder, P, p, batch, None)
    if decoder == 'disc':
        return disc_decoder_forward(P, p, batch['options'], enc_out)[0]
    return gen_retrieve_scores(P, p, batch, enc_out, st)


def forward_backward(encoder, decoder, P, p, batch, drop=None, only_forward=False):
    """Model:forwardBackward (model.lua:249-342).  Returns dict(loss, scores, grads, enc_out)."""
    P = dict(P)
    P['embed'] = P['embed'].copy()
    P['embed'][0] = 0          # LookupTableMaskZero zeroes the pad row on every forward
    enc_out, st = encoder_forward(encoder, P, p, batch, drop)
    out = dict(enc_out=enc_out, grads=None, scores=None)
    if decoder == 'disc':
        scores, dst = disc_decoder_forward(P, p, batch['options'], enc_out)
        loss, dscores, loss_rows = cross_entropy(scores, batch['answer_ind'] - 1)   # answer_ind is 1-based (prepro.py:169)
        out.update(loss=loss, scores=scores, loss_rows=loss_rows)
        if only_forward:
            return out
        G = {k: np.zeros_like(v) for k, v in P.items()}
        denc = disc_decoder_backward(P, G, dst, enc_out, dscores)
        encoder_backward(encoder, P, G, p, batch, drop, st, denc)
    elif decoder == 'gen':
        loss, gst = gen_decoder_forward(P, p, batch, enc_out, st)
        out.update(loss=loss)
        if only_forward:
            return out
        G = {k: np.zeros_like(v) for k, v in P.items()}
        denc, dh0s, dc0s = gen_decoder_backward(P, G, p, gst)
        encoder_backward(encoder, P, G, p, batch, drop, st, denc, dh0s, dc0s)
    else:
        raise ValueError(decoder)
    out['grads'] = G
    return out


def train_iteration(encoder, decoder, P, p, batch, drop, opt_state, lr):
    """Model:trainIteration (model.lua:66-106) on explicit batch: fwd/bwd, clamp +-5, adam."""
    spec = param_spec(encoder, decoder, p)
    r = forward_backward(encoder, decoder, P, p, batch, drop)
    # LookupTableMaskZero zeroes its pad row IN the flat parameter vector on every forward: the update starts from that row = 0 (its
    # gradient need not be zero: where the embedding is joined with another input before the LSTM, pad steps are not masked)
    P = dict(P)
    P['embed'] = P['embed'].copy()
    P['embed'][0] = 0
    w = flatten(P, spec)
    g = flatten(r['grads'], spec)
    w2, g2 = clamp_adam(w, g, opt_state, lr)
    return unflatten(w2, spec), r
