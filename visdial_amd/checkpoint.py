"""Checkpoint I/O: this repo's own `.pt` files (torch.save of {modelW, optims, modelParams}, same three
fields as train.lua:99-102) and the reference's Torch7 `.t7` files (via visdial_amd.t7; the flat vector is in the
reference's getParameters() order for every encoder -- t7.reference_order, derived by executing the reference's files)."""
import sys

import numpy as np
import torch

from . import t7


def load_checkpoint(path):
    if path.endswith('.t7'):
        ck = t7.load(path)
        ck['_flat_reference_layout'] = True
        return ck
    return torch.load(path, weights_only=True)      # plain tensors and primitives only


def restore_weights(model, saved, param_order=None):
    """`model.wrapperW:copy(savedModel.modelW)` (train.lua:79, evaluate.lua:91, generate.lua:83).  A Torch7-format flat vector is split
    in the reference's getParameters() order (t7.reference_order); for the four nngraph encoders that order is DERIVED
    (t7.order_status) and the loader says so on stderr.  `param_order` (-paramOrder) overrides it: 'declaration', 'reference', or a JSON
    file -- an explicit name list or the output of lua/dump_param_order.lua under a real Torch7 (t7.resolve_order).  The element count
    is always checked; the forget-bias pattern of a fresh reference initialisation is reported when present."""
    w = saved['modelW']
    if saved.get('_flat_reference_layout'):
        note = lambda m: sys.stderr.write('note: ' + m + '\n')
        enc = model.params['encoder']
        entries = model._entries() if hasattr(model, '_entries') else model.fp.spec.entries
        marker = saved.get('vdLayout')
        order = param_order or ('declaration' if marker == 'declaration' else None)
        if not order and t7.order_status(enc) == 'derived':
            note("'%s' is an nngraph encoder; its getParameters() order is derived (reference file executed on a restated nngraph), not "
                 "verified against a Torch7-written checkpoint -- `th lua/dump_param_order.lua <ckpt>` + -paramOrder <json> checks it" % enc)
        if not order and marker is None and (enc.startswith('hre') or t7.order_status(enc) == 'derived'):
            note("the file carries no vdLayout marker: read as a Torch7 / reference-order vector.  (A file written through this repo's "
                 "Lua host BEFORE round 4 is in declaration order: pass -paramOrder declaration.)")
        named = t7.flat_to_named(np.asarray(w, np.float32), entries, enc, order, note)
        n, ok = t7.forget_bias_report(named, entries)
        if n and ok == n:
            note("all %d LSTM biases carry the forget-gate-bias-1 pattern of a fresh reference initialisation: the split is plausible" % n)
        elif n and ok:
            note("%d of %d LSTM biases carry the fresh-initialisation forget-bias pattern, the others do not: CHECK the parameter order" % (ok, n))
        model.set_parameters_dict(named)
    else:
        model.wrapperW.copy_(w.to(model.wrapperW.device))
        # the copy runs on torch's current stream; the native library's streams are non-blocking and never order
        # against it -- finish it before the next library call can read the weights
        torch.cuda.current_stream().synchronize()


def save_t7(path, model, params, optims=True):
    """torch.save(path, {modelW = ..., optims = ..., modelParams = ...}) in the Torch7 binary format (train.lua:99-102,
    120-121; FloatTensor flavour like convert_gpu_to_cpu.lua:30-45).  The flat vector is in the reference's
    getParameters() order for this encoder (t7.reference_order) -- the same order lua/model.lua:Model:tensors() uses, so a
    file written by either host loads in the other and, as far as the derived order holds, in Torch7."""
    clean = {k: v for k, v in params.items() if isinstance(v, (int, float, str, bool)) or v is None}
    enc = params['encoder']
    entries = model._entries() if hasattr(model, '_entries') else model.fp.spec.entries
    named = model.get_parameters_dict()
    obj = {'modelW': t7.named_to_flat(named, entries, enc), 'modelParams': clean, 'vdLayout': 'reference'}
    if optims:
        obj['optims'] = {'learningRate': float(model.optims['learningRate'])}
    t7.save(path, obj)
