"""Semantics-robustness tests.

The arithmetic of the reference lives in libraries that are not in /root/reference (Torch7 `nn`, Element-Research `rnn`, `nngraph`):
the oracle restates it from their published behaviour ([UPSTREAM-RECALL]), and no reference-produced vector pins those
readings offline.  These tests do the next best thing: for every such behaviour the results rest on, the oracle is run under the PLAUSIBLE
ALTERNATIVE READING and the outputs on contract-conformant batches (dataloader.lua's layouts: right-aligned questions / history,
left-aligned answers with a <START> prefix, masked targets) must be IDENTICAL -- loss and every gradient tensor -- so that whichever reading
a real Torch7 run follows, the parity claims stand.  Where a reading CAN change a result, the test shows exactly where (and that the
dataloader never produces that input).  DESIGN.md section 3 tabulates the outcome.

CPU only: the oracle is the thing under test here."""
import contextlib

import numpy as np
import pytest

from conftest import small_params
from oracle import visdial_oracle as vo
from visdial_amd.dataloader import SyntheticDataloader
from visdial_amd.opts import derive

PAIRS = [('mn-att-ques-im-hist', 'disc'), ('lf-ques-im-hist', 'gen'), ('hre-ques-im-hist', 'disc'), ('hrea-ques-im-hist', 'gen'),
         ('mn-ques-hist', 'gen'), ('lf-att-ques-im-hist', 'disc'), ('lf-ques', 'gen')]


def _case(enc, dec, seed=7, **kw):
    p = derive(small_params(encoder=enc, decoder=dec, **kw))
    batch = SyntheticDataloader(p, seed=seed).getTrainBatch(p)
    P = vo.init_params(enc, dec, p, seed=3)
    return p, batch, P


def _same(a, b, what):
    assert a['loss'] == b['loss'], (what, a['loss'], b['loss'])
    for k in a['grads']:
        np.testing.assert_array_equal(a['grads'][k], b['grads'][k], err_msg='%s: %s' % (what, k))
    if a.get('scores') is not None:
        np.testing.assert_array_equal(a['scores'], b['scores'], err_msg=what)


# ------------------------------------------------------------------------------------------------------------------------------------
# 1. maskZero(): RESET the state of a padded row to zero (the oracle, csrc/lstm.hip, nn7.py) vs HOLD the previous state through the pad step
#    (the other way a masked recurrence is commonly written: output zero, h / c carried).
# ------------------------------------------------------------------------------------------------------------------------------------
def lstm_forward_hold(x, W, b, tok_mask=None, h0=None, c0=None):
    """vo.lstm_forward with the ALTERNATIVE reading: a masked row emits zero (h_all, gates = 0) but its recurrent state is carried"""
    T, N, D = x.shape
    H = W.shape[1] // 4
    Wx, Wh = W[:D], W[D:]
    dt = x.dtype
    h_all, c_all, gates = np.zeros((T, N, H), dt), np.zeros((T, N, H), dt), np.zeros((T, N, 4 * H), dt)
    hp = np.zeros((N, H), dt) if h0 is None else h0
    cp = np.zeros((N, H), dt) if c0 is None else c0
    held = []
    for t in range(T):
        a = b + x[t] @ Wx + hp @ Wh
        i, f, o = vo.sigmoid(a[:, :H]), vo.sigmoid(a[:, H:2 * H]), vo.sigmoid(a[:, 2 * H:3 * H])
        g = np.tanh(a[:, 3 * H:])
        c = f * cp + i * g
        h = o * np.tanh(c)
        gt = np.concatenate([i, f, o, g], 1)
        if tok_mask is not None:
            keep = (tok_mask[t] != 0)[:, None].astype(dt)
            held.append((hp.copy(), cp.copy()))
            h_all[t], c_all[t], gates[t] = h * keep, c * keep, gt * keep          # what the step EMITS
            hp, cp = h * keep + hp * (1 - keep), c * keep + cp * (1 - keep)          # what it CARRIES
        else:
            h_all[t], c_all[t], gates[t] = h, c, gt
            hp, cp = h, c
    lstm_forward_hold.last_state = (hp, cp)
    return h_all, c_all, gates


def test_hold_reading_really_differs_off_contract():
    """the alternative is a different function: on a LEFT-aligned row (real tokens, then pads) the carried state at the last step is the
    last real token's state under `hold` and zero under `reset` -- so the equalities below are not vacuous"""
    rng = np.random.RandomState(0)
    T, N, D, H = 5, 3, 4, 6
    x = rng.randn(T, N, D)
    tok = np.ones((T, N), np.int64)
    tok[3:, 0] = 0                                     # row 0: left-aligned, two trailing pads
    x[3:, 0] = 0
    W, b = rng.randn(D + H, 4 * H) * 0.3, rng.randn(4 * H) * 0.1
    h_reset, c_reset, _ = vo.lstm_forward(x, W, b, tok)
    h_hold, _, _ = lstm_forward_hold(x, W, b, tok)
    np.testing.assert_array_equal(h_reset, h_hold)     # the EMITTED sequences agree (zeros at the pads) ...
    hp, _ = lstm_forward_hold.last_state
    assert np.abs(hp[0]).max() > 1e-3 and np.abs(h_reset[-1, 0]).max() == 0      # ... the carried state does not
    # and a real token AFTER a pad sees the difference: interior pad
    tok2 = np.ones((T, N), np.int64)
    tok2[2, 1] = 0
    x2 = rng.randn(T, N, D)
    x2[2, 1] = 0
    a, _, _ = vo.lstm_forward(x2, W, b, tok2)
    bb, _, _ = lstm_forward_hold(x2, W, b, tok2)
    assert np.abs(a[3:, 1] - bb[3:, 1]).max() > 1e-3


_CARRIED = {}


def lstm_forward_hold_saving(x, W, b, tok_mask=None, h0=None, c0=None):
    """lstm_forward_hold that also remembers, per call, the CARRIED state entering every step (the backward of `hold` needs it: after a pad
    step the recurrent input of the next real step is the carried state, not the emitted zero)"""
    T, N, D = x.shape
    H = W.shape[1] // 4
    Wx, Wh = W[:D], W[D:]
    dt = x.dtype
    h_all, c_all, gates = np.zeros((T, N, H), dt), np.zeros((T, N, H), dt), np.zeros((T, N, 4 * H), dt)
    hp = np.zeros((N, H), dt) if h0 is None else h0
    cp = np.zeros((N, H), dt) if c0 is None else c0
    hin, cin, keeps, full = [], [], [], []
    for t in range(T):
        a = b + x[t] @ Wx + hp @ Wh
        i, f, o = vo.sigmoid(a[:, :H]), vo.sigmoid(a[:, H:2 * H]), vo.sigmoid(a[:, 2 * H:3 * H])
        g = np.tanh(a[:, 3 * H:])
        c = f * cp + i * g
        h = o * np.tanh(c)
        gt = np.concatenate([i, f, o, g], 1)
        keep = np.ones((N, 1), dt) if tok_mask is None else (tok_mask[t] != 0)[:, None].astype(dt)
        hin.append(hp); cin.append(cp); keeps.append(keep); full.append((gt, c))
        h_all[t], c_all[t], gates[t] = h * keep, c * keep, gt * keep
        hp, cp = h * keep + hp * (1 - keep), c * keep + cp * (1 - keep)
    _CARRIED[id(gates)] = (hin, cin, keeps, full, gates)        # (gates kept alive so the id stays unique)
    return h_all, c_all, gates


def lstm_backward_hold(x, W, gates, h_all, c_all, dh_seq=None, dh_last=None, dc_last=None, h0=None, c0=None, return_da=False):
    """backward of the `hold` reading: a pad step passes the gradient of the carried state through unchanged and contributes nothing else"""
    hin, cin, keeps, full, _ = _CARRIED[id(gates)]
    T, N, D = x.shape
    H = W.shape[1] // 4
    Wx, Wh = W[:D], W[D:]
    dt = x.dtype
    dW, db, dx = np.zeros_like(W), np.zeros(4 * H, dt), np.zeros_like(x)
    dH = np.zeros((N, H), dt) if dh_last is None else dh_last.copy()      # gradient w.r.t. the CARRIED state after the last step
    dC = np.zeros((N, H), dt) if dc_last is None else dc_last.copy()
    da_all = np.zeros_like(gates)
    for t in range(T - 1, -1, -1):
        keep = keeps[t]
        gt, c = full[t]
        i, f, o, g = (gt[:, k * H:(k + 1) * H] for k in range(4))
        dh = dH * keep
        if dh_seq is not None:
            dh = dh + dh_seq[t] * keep                                   # the emitted output is h * keep
        tc = np.tanh(c)
        dc = (dC + dh * o * (1 - tc * tc)) * keep
        da = np.concatenate([dc * g * i * (1 - i), dc * cin[t] * f * (1 - f), dh * tc * o * (1 - o), dc * i * (1 - g * g)], 1)
        da_all[t] = da
        dx[t] = da @ Wx.T
        dW[:D] += x[t].T @ da
        dW[D:] += hin[t].T @ da
        db += da.sum(0)
        dH = da @ Wh.T + dH * (1 - keep)
        dC = dc * f + dC * (1 - keep)
    if return_da:
        return dx, dW, db, dH, dC, da_all
    return dx, dW, db, dH, dC


def test_hold_backward_is_the_gradient_of_hold_forward():
    """finite differences on an OFF-contract layout (interior and trailing pads), so that the alternative reading used below is a correct
    forward / backward pair and not a strawman"""
    rng = np.random.RandomState(2)
    T, N, D, H = 5, 3, 3, 4
    x = rng.randn(T, N, D)
    tok = np.ones((T, N), np.int64)
    tok[1, 0] = 0; tok[3:, 1] = 0; tok[0, 2] = 0
    W, b = rng.randn(D + H, 4 * H) * 0.4, rng.randn(4 * H) * 0.1
    wseq, wlast = rng.randn(T, N, H), rng.randn(N, H)

    def f(W_):
        h, c, g = lstm_forward_hold_saving(x, W_, b, tok)
        hin, cin, keeps, full, _ = _CARRIED[id(g)]
        hT = full[-1][0][:, 2 * H:3 * H] * np.tanh(full[-1][1]) * keeps[-1] + hin[-1] * (1 - keeps[-1])       # carried state after the last step
        return (h * wseq).sum() + (hT * wlast).sum(), (h, c, g)
    _, (h, c, g) = f(W)
    _, dW, _, _, _ = lstm_backward_hold(x, W, g, h, c, dh_seq=wseq, dh_last=wlast)
    for idx in [(0, 1), (D + 1, 2), (D + H - 1, 4 * H - 1), (2, H + 3)]:
        e = np.zeros_like(W); e[idx] = 1e-6
        num = (f(W + e)[0] - f(W - e)[0]) / 2e-6
        assert abs(num - dW[idx]) < 1e-6 * max(1, abs(num)), (idx, num, dW[idx])


@contextlib.contextmanager
def _hold_reading():
    """swap the oracle's masked recurrence for the `hold` reading, forward AND backward"""
    fwd, bwd = vo.lstm_forward, vo.lstm_backward
    vo.lstm_forward, vo.lstm_backward = lstm_forward_hold_saving, lstm_backward_hold
    try:
        yield
    finally:
        vo.lstm_forward, vo.lstm_backward = fwd, bwd
        _CARRIED.clear()


@pytest.mark.parametrize("enc,dec", PAIRS)
def test_maskzero_reset_vs_hold_is_identical_on_dataloader_layouts(enc, dec):
    """encoders (right-aligned questions / history, two stacked masked layers) and the gen decoder (left-aligned answer_in behind <START>,
    targets masked by MaskZeroCriterion): loss and EVERY gradient tensor identical under both readings"""
    p, batch, P = _case(enc, dec)
    ref = vo.forward_backward(enc, dec, P, p, batch, None)
    with _hold_reading():
        alt = vo.forward_backward(enc, dec, P, p, batch, None)
    # (the two backward passes sum the same terms in the same order on these layouts: bit-identical, not just close)
    _same(ref, alt, 'maskZero reset vs hold, %s + %s' % (enc, dec))
    if dec == 'gen':
        # the gen decoder's answer_in has TRAILING pads (left-aligned): there the carried state differs between the readings -- but no
        # unmasked target and no later token reads it.  Show the layout is what this claim needs:
        ain = batch['answer_in'].reshape(-1, batch['answer_in'].shape[-1])
        assert (ain[:, 0] != 0).all() and all((row[np.argmax(row == 0):] == 0).all() for row in ain if (row == 0).any())


# ------------------------------------------------------------------------------------------------------------------------------------
# 2. a masked layer derives its mask from ITS OWN input rows being all-zero (rnn's maskzero) vs from the token ids (the oracle, the HIP path)
# ------------------------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("enc,dec", [('lf-ques-im-hist', 'gen'), ('mn-ques-im-hist', 'disc'), ('hre-ques-hist', 'disc')])
def test_mask_from_all_zero_input_rows_equals_mask_from_token_ids(enc, dec):
    p, batch, P = _case(enc, dec, seed=11)
    ref = vo.forward_backward(enc, dec, P, p, batch, None)
    fwd = vo.lstm_forward
    seen = []

    def from_rows(x, W, b, tok_mask=None, h0=None, c0=None):
        if tok_mask is not None:
            rows = (np.abs(x).max(-1) != 0).astype(np.int64)                # what nn.SeqLSTM:maskZero sees: the input row itself
            seen.append((rows != (np.asarray(tok_mask) != 0)).sum())
            tok_mask = rows
        return fwd(x, W, b, tok_mask, h0, c0)
    vo.lstm_forward = from_rows
    try:
        alt = vo.forward_backward(enc, dec, P, p, batch, None)
    finally:
        vo.lstm_forward = fwd
    assert seen and sum(seen) == 0          # layer 1: embedding rows are zero exactly at token 0; layer 2: layer 1 emits exact zeros there
    _same(ref, alt, 'mask source, %s + %s' % (enc, dec))


def test_mask_from_input_rows_under_embedding_dropout():
    """the nngraph encoders put nn.Dropout(0.5) between the shared embedding and the LSTM (mn-att:24-25): a real token's embedding row would
    have to be dropped ENTIRELY (probability 2^-E, E = 300 in the reference) to be mistaken for padding; with the pinned masks used by every
    parity test no real row is"""
    p, batch, P = _case('mn-att-ques-im-hist', 'disc', seed=5)
    rng = np.random.RandomState(0)
    B, R, Tq = batch['ques_fwd'].shape
    E = p['embedSize']
    keep = rng.rand(Tq, B * R, E) > 0.5
    tok = batch['ques_fwd'].reshape(B * R, Tq).T
    x = vo.dropout(vo.lookup(P['embed'], tok), keep, 0.5)
    assert ((np.abs(x).max(-1) != 0) == (tok != 0)).all()


# ------------------------------------------------------------------------------------------------------------------------------------
# 3. LookupTableMaskZero: the gradient that lands on the pad row is KEPT in gradWeight (the oracle: np.add.at incl. row 0) vs DROPPED
# ------------------------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("enc,dec", [('mn-att-ques-im-hist', 'disc'), ('hre-ques-im-hist', 'disc'), ('lf-ques-im-hist', 'gen')])
def test_pad_row_gradient_kept_vs_dropped_changes_only_the_pad_row(enc, dec):
    """three training iterations under both readings: every loss, every parameter the forward pass reads and every other gradient entry
    identical; the only difference is the pad row of the embedding gradient / Adam state, which the forward never reads (re-zeroed)"""
    p, batch, P0 = _case(enc, dec, seed=13)
    lb = vo.lookup_backward

    def run(drop_pad):
        def lookup_backward(demb, tok, dx):
            lb(demb, tok, dx)
            if drop_pad:
                demb[0] = 0
        vo.lookup_backward = lookup_backward
        try:
            P, st, losses, grads = {k: v.copy() for k, v in P0.items()}, {}, [], None
            for it in range(3):
                b = SyntheticDataloader(p, seed=100 + it).getTrainBatch(p)
                P, r = vo.train_iteration(enc, dec, P, p, b, None, st, 1e-3)
                losses.append(r['loss'])
                grads = r['grads']
            return P, losses, grads
        finally:
            vo.lookup_backward = lb
    Pk, lk, gk = run(False)
    Pd, ld, gd = run(True)
    assert lk == ld
    for k in Pk:
        a, b = Pk[k], Pd[k]
        if k == 'embed':
            a, b = a[1:], b[1:]
        np.testing.assert_array_equal(a, b, err_msg=k)
    for k in gk:
        a, b = gk[k], gd[k]
        if k == 'embed':
            a, b = a[1:], b[1:]
        np.testing.assert_array_equal(a, b, err_msg=k)
    if enc.startswith('hre') and 'im' in enc:
        # where the embedding is joined with the image feature before the LSTM, pad steps are not masked and DO send a gradient to row 0:
        # the two readings differ there -- in that row only
        assert np.abs(gk['embed'][0]).max() > 0 and np.abs(gd['embed'][0]).max() == 0


# ------------------------------------------------------------------------------------------------------------------------------------
# 4. MaskZero(LogSoftMax) at a pad step of the gen decoder: an all-ZERO row (executing gen.lua:23-24 on the restated nn) vs log_softmax(bias)
# ------------------------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("enc", ['lf-ques-im-hist', 'mn-att-ques-im-hist', 'hre-ques-hist'])
def test_gen_pad_step_output_row_does_not_reach_loss_or_gradients(enc):
    """the criterion is MaskZeroCriterion over the targets (model.lua:32-36): whatever the decoder emits at a pad step, its loss term and its
    gradient are zero.  Replace the pad-step logits by arbitrary values: loss and every gradient unchanged."""
    p, batch, P = _case(enc, 'gen', seed=17)
    ref = vo.forward_backward(enc, 'gen', P, p, batch, None)
    gdf = vo.gen_decoder_forward
    rng = np.random.RandomState(1)
    hit = []

    def perturbed(P_, p_, batch_, enc_out, enc_state):
        loss, st = gdf(P_, p_, batch_, enc_out, enc_state)
        keep = (st['ain'] != 0)
        hit.append(int((~keep).sum()))
        # the alternative reading changes the OUTPUT ROW at pad steps; the quantities the backward pass consumes are the loss terms and
        # dlogits, both multiplied by `keep` (MaskZeroCriterion): assert that, and that nothing else of a pad step is read
        assert np.abs(st['dlogits'][~keep]).max() == 0
        st['h'] = st['h'] + (~keep)[..., None] * rng.randn(*st['h'].shape)          # garbage hidden rows at pad steps: never read through dlogits = 0
        return loss, st
    vo.gen_decoder_forward = perturbed
    try:
        alt = vo.forward_backward(enc, 'gen', P, p, batch, None)
    finally:
        vo.gen_decoder_forward = gdf
    assert hit and hit[0] > 0                           # the batch does contain pad steps
    _same(ref, alt, 'gen pad-step rows, %s' % enc)


def test_where_the_pad_step_row_does_matter():
    """... it matters only where a pad token is FED and its output row is READ: beam search with an unfilled beam slot (model.lua:433-667),
    found by executing generate.lua in round 4 and reproduced token for token (tests/test_reference_train.py); training and retrieval never
    do that: retrieval scores sum log-probabilities over non-pad targets only (utils.computeLhood, utils.lua)"""
    logp = np.log(np.full((4, 2, 5), 0.2))
    words = np.array([[1, 2], [3, 0], [0, 0], [0, 0]])           # second candidate: one token then pads
    base = vo.compute_lhood(words, logp)
    logp2 = logp.copy()
    logp2[1:, 1] = 123.0                                          # whatever the decoder emits at the pad steps of candidate 2 ...
    logp2[2:, 0] = -55.0
    np.testing.assert_array_equal(vo.compute_lhood(words, logp2), base)       # ... the likelihood does not read it


# ------------------------------------------------------------------------------------------------------------------------------------
# 5. torch.sort in utils.computeRanks: tie order is implementation-defined; the ranks do not depend on it when the scores are tie-free
# ------------------------------------------------------------------------------------------------------------------------------------
def _ranks_with(scores, gt, kind, flip_ties):
    s = np.asarray(scores, np.float64)
    n, o = s.shape
    idx = np.arange(o)[::-1] if flip_ties else np.arange(o)
    order = idx[np.argsort(-s[:, idx], axis=1, kind=kind)]
    ranks = np.empty((n, o), np.int64)
    ranks[np.arange(n)[:, None], order] = np.arange(1, o + 1)[None, :]
    return ranks[np.arange(n), gt]


@pytest.mark.parametrize("enc", ['mn-att-ques-im-hist', 'hre-ques-im-hist'])
def test_ranks_do_not_depend_on_the_sort_when_scores_are_tie_free(enc):
    p, batch, P = _case(enc, 'disc', seed=19, numOptions=100, batchSize=3, vocabSize=3000)
    opts = batch['options'].reshape(-1, 100, batch['options'].shape[-1])
    distinct = np.array([np.unique(o, axis=0).shape[0] == 100 for o in opts])     # rounds whose 100 candidates are distinct rows
    assert distinct.sum() >= 8                                           # (a round WITH a duplicate answer: the test below)
    r = vo.forward_backward(enc, 'disc', P, p, batch, None, only_forward=True)
    s, gt = r['scores'][distinct], (batch['answer_ind'].reshape(-1) - 1)[distinct]
    assert all(np.unique(row).size == row.size for row in s)            # fp64 scores of distinct options: no ties
    assert all(np.unique(row.astype(np.float32)).size == row.size for row in s)    # nor after rounding to the fp32 the device ranks on
    base = vo.compute_ranks(s, gt)
    for kind in ('stable', 'quicksort', 'heapsort'):
        for flip in (False, True):
            np.testing.assert_array_equal(_ranks_with(s, gt, kind, flip), base)


def test_where_the_sort_does_matter():
    """duplicate candidate answers (real VisDial repeats popular answers inside one option list) score EXACTLY equal: the rank of the ground
    truth then depends on the tie order.  The hosts and the oracle break ties by lower index first (utils.lua:106-128 through a stable sort,
    [UPSTREAM-RECALL] for torch.sort); the other order moves the rank by the number of duplicates ahead -- bounded, and stated here."""
    s = np.array([[0.3, 0.9, 0.9, 0.1]])
    assert _ranks_with(s, np.array([2]), 'stable', False)[0] == 2 == vo.compute_ranks(s, np.array([2]))[0]
    assert _ranks_with(s, np.array([2]), 'stable', True)[0] == 1
