"""CPU ORACLE (test infrastructure only) for the data path: a loop-by-loop restatement of the reference's
dataloader.lua preprocessing and batch assembly and of utils.rightAlign, written independently of the
vectorised product code in visdial_amd/dataloader.py.  PARITY UNPINNED (no Lua runtime, no fixtures in the
reference); every function cites the lines it follows.  Indices here are 0-based; comments give Lua lines."""
import numpy as np


def right_align(seq, lengths):
    """utils.lua:6-45"""
    out = np.zeros_like(seq)
    M = seq.shape[-1]
    if seq.ndim == 3:
        for i in range(seq.shape[0]):
            for q in range(seq.shape[1]):
                L = int(lengths[i][q])
                if L == 0:
                    break                                  # utils.lua:21-23
                out[i, q, M - L:] = seq[i, q, :L]
    else:
        for i in range(seq.shape[0]):
            L = int(lengths[i])
            if L > 0:
                out[i, M - L:] = seq[i, :L]
    return out


def process_answers(ans, ans_len, START, END):
    """dataloader.lua:159-200"""
    n, R, M = ans.shape
    din = np.zeros((n, R, M + 1), np.int64)
    dout = np.zeros((n, R, M + 1), np.int64)
    din[:, :, 0] = START
    for i in range(n):
        for r in range(R):
            L = int(ans_len[i, r])
            if L > 0:
                din[i, r, 1:L + 1] = ans[i, r, :L]
                dout[i, r, :L] = ans[i, r, :L]
            dout[i, r, L] = END
    return din, dout, ans_len + 1


def process_options(opt_list, opt_len, max_ans_len, START, END):
    """dataloader.lua:281-321"""
    n = opt_list.shape[0]
    din = np.zeros((n, max_ans_len + 1), np.int64)
    dout = np.zeros((n, max_ans_len + 1), np.int64)
    din[:, 0] = START
    for i in range(n):
        L = int(opt_len[i])
        if L > 0:
            din[i, 1:L + 1] = opt_list[i, :L]
            dout[i, :L] = opt_list[i, :L]
            dout[i, L] = END
    return din, dout, opt_len + 1


def process_history(cap, cap_len, ques, ques_len, ans, ans_len, concat, END):
    """dataloader.lua:203-278 (ans_len BEFORE processAnswers increments it)"""
    n, R, MQ = ques.shape
    MA = ans.shape[2]
    W0 = MQ + MA
    W = min(R * W0, 300) if concat else W0
    hist = np.zeros((n, R, W), np.int64)
    hl = np.zeros((n, R), np.int64)
    for i in range(n):
        lenH = 0
        for r in range(R):
            if r == 0:
                hist[i, 0, :W0] = cap[i, :W0]
                lenH = min(int(cap_len[i]), W0)
            else:
                lq, la = int(ques_len[i, r - 1]), int(ans_len[i, r - 1])
                if concat:
                    for k in range(lenH):
                        hist[i, r, k] = hist[i, r - 1, k]
                    hist[i, r, lenH] = END
                    for k in range(lq):
                        hist[i, r, lenH + 1 + k] = ques[i, r - 1, k]
                    for k in range(la):
                        hist[i, r, lenH + 1 + lq + k] = ans[i, r - 1, k]
                    lenH = lenH + lq + la + 1
                else:
                    for k in range(lq):
                        hist[i, r, k] = ques[i, r - 1, k]
                    for k in range(la):
                        hist[i, r, lq + k] = ans[i, r - 1, k]
                    lenH = lq + la
            hl[i, r] = lenH
    return right_align(hist, hl), hl, W


def index_data(d, inds0, use_hist, max_history_len, test=False):
    """dataloader.lua:378-432 for 0-based thread ids"""
    out = {}
    mq = max(int(d['ques_len'][i].max()) for i in inds0)
    out['ques_fwd'] = np.stack([d['ques_fwd'][i][:, d['ques_fwd'].shape[2] - mq:] for i in inds0])
    if use_hist:
        mh = min(max(int(d['hist_len'][i].max()) for i in inds0), max_history_len)
        out['hist'] = np.stack([d['hist'][i][:, d['hist'].shape[2] - mh:] for i in inds0])
    ma = max(int(d['ans_len1'][i].max()) for i in inds0)
    out['answer_in'] = np.stack([d['ans_in'][i][:, :ma] for i in inds0])
    out['answer_out'] = np.stack([d['ans_out'][i][:, :ma] for i in inds0])
    if not test:
        out['answer_ind'] = np.stack([d['ans_ind'][i] for i in inds0])
    return out
