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


def restore_weights(model, saved):
    """`model.wrapperW:copy(savedModel.modelW)` (train.lua:79, evaluate.lua:91, generate.lua:83).  A Torch7-written flat
    vector is split in the reference's getParameters() order (t7.reference_order); for the four nngraph encoders that
    order is DERIVED (t7.order_status) and the loader says so on stderr.  The element count is always checked."""
    w = saved['modelW']
    if saved.get('_flat_reference_layout'):
        if saved.get('vdLayout') == 'declaration':
            # written by an earlier save_t7 of THIS repo for an nngraph encoder: tensors back to back in the library's
            # own declaration order (vd_model_tensor_info)
            from . import t7 as _t7
            entries = model._entries() if hasattr(model, '_entries') else model.fp.spec.entries
            model.set_parameters_dict(_t7.flat_to_named(np.asarray(w, np.float32), entries, None))
        else:
            enc = model.params['encoder']
            if t7.order_status(enc) == 'derived':
                sys.stderr.write("note: '%s' is an nngraph encoder; its getParameters() order is derived (reference file executed on "
                                 "a restated nngraph), not verified against a Torch7-written checkpoint\n" % enc)
            model.load_flat_parameters(np.asarray(w, np.float32))
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
    obj = {'modelW': t7.named_to_flat(named, entries, enc), 'modelParams': clean}
    if optims:
        obj['optims'] = {'learningRate': float(model.optims['learningRate'])}
    t7.save(path, obj)
