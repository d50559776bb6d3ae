"""Counterparts of the reference's utils.lua ranking helpers (utils.lua:106-160)."""
import json

import numpy as np
import torch

from . import ops


def computeRanks(scores, gtPos=None, ws=None):
    """utils.lua:106-128.  scores: device [N x O]; gtPos: device int32 [N] 0-based or None.
    The reference sorts on the device and converts positions to ranks in a host Lua loop; here the
    rank of every option is one counting kernel (K10).  Returns float64 numpy like `ranks:double()`."""
    N, O = scores.shape
    out = ws.get('ranks.out', (N, O), torch.int32) if ws is not None else torch.empty(N, O, dtype=torch.int32,
                                                                                       device=scores.device)
    ops.ranks(scores, out, N, O)
    ranks = out.cpu().numpy()
    if gtPos is not None:
        g = gtPos.cpu().numpy().reshape(-1)
        ranks = ranks[np.arange(N), g]
    return ranks.astype(np.float64)


def processRanks(ranks, verbose=True):
    """utils.lua:131-160 (incl. its quirk of dividing by the unfiltered question count)."""
    ranks = np.asarray(ranks, np.float64)
    numQues = ranks.size
    numOptions = 100
    r = ranks.reshape(-1)
    if (r <= 0).sum() > 0:
        if verbose:
            print('Warning: some of ranks are zero : %d' % int((r <= 0).sum()))
        r = r[r > 0]
    if (r >= numOptions + 1).sum() > 0:
        if verbose:
            print('Warning: some of ranks >100 : %d' % int((r >= numOptions + 1).sum()))
        r = r[r <= numOptions + 1]
    m = {
        'numQues': numQues,
        'r@1': float((r <= 1).sum()) / numQues,
        'r@5': float((r <= 5).sum()) / numQues,
        'r@10': float((r <= 10).sum()) / numQues,
        'medianR': float(np.sort(r)[(r.size - 1) // 2]) if r.size else 0.0,   # torch.median: lower middle
        'meanR': float(r.mean()) if r.size else 0.0,
        'meanRR': float((1.0 / r).mean()) if r.size else 0.0,
    }
    if verbose:
        print('\tNo. questions: %d' % m['numQues'])
        for k in ('r@1', 'r@5', 'r@10', 'medianR', 'meanR', 'meanRR'):
            print('\t%s: %f' % (k, m[k]))
    return m


def writeJSON(path, obj):
    """utils.lua:76-83"""
    with open(path, 'w') as f:
        json.dump(obj, f)


def idToWords(vector, ind2word):
    """utils.lua:48-63: ' w1 w2 ...' up to and including <END>; zeros are skipped."""
    sentence = ''
    nextWord = None
    for w in list(vector):
        w = int(w)
        if w > 0:
            nextWord = ind2word[w]
            sentence = sentence + ' ' + nextWord
        if nextWord == '<END>':
            break
    return sentence
