"""Checkpoint I/O: this repo's own `.pt` files (torch.save of {modelW, optims, modelParams}, same three
fields as train.lua:99-102) and the reference's Torch7 `.t7` files (via visdial_amd.t7; flat-vector order
verified only for the documented Sequential-encoder layout -- see t7.flat_to_named)."""
import numpy as np
import torch

from . import t7


def load_checkpoint(path):
    if path.endswith('.t7'):
        ck = t7.load(path)
        ck['_flat_reference_layout'] = True
        return ck
    return torch.load(path, weights_only=True)      # plain tensors and primitives only


def restore_weights(model, saved, allow_unverified=False):
    """`model.wrapperW:copy(savedModel.modelW)` (train.lua:79, evaluate.lua:91, generate.lua:83).  allow_unverified
    (CLI flag -allowUnverifiedOrder 1) accepts a Torch7-written flat vector for the four nngraph encoders, whose
    getParameters() order cannot be verified offline (t7.flat_to_named)."""
    w = saved['modelW']
    if saved.get('_flat_reference_layout'):
        if saved.get('vdLayout') == 'declaration':
            # written by save_t7 of THIS repo for an nngraph encoder: tensors back to back in the library's own
            # declaration order (vd_model_tensor_info) -- no guess about nngraph's node order is involved
            from . import t7 as _t7
            entries = model._entries() if hasattr(model, '_entries') else model.fp.spec.entries
            model.set_parameters_dict(_t7.flat_to_named(np.asarray(w, np.float32), entries, None))
        else:
            model.load_flat_parameters(np.asarray(w, np.float32), allow_unverified=allow_unverified)
    else:
        model.wrapperW.copy_(w.to(model.wrapperW.device))
        # the copy runs on torch's current stream; the native library's streams are non-blocking and never order
        # against it -- finish it before the next library call can read the weights
        torch.cuda.current_stream().synchronize()


def save_t7(path, model, params, optims=True):
    """torch.save(path, {modelW = ..., optims = ..., modelParams = ...}) in the Torch7 binary format (train.lua:99-102,
    120-121; FloatTensor flavour like convert_gpu_to_cpu.lua:30-45).  For the Sequential-built encoders the flat
    vector is in the reference's getParameters() order (t7.VERIFIED_ORDER); for the four nngraph encoders that order
    is not derivable offline, so the vector is written in this library's declaration order and the table says so
    (`vdLayout = 'declaration'`) -- restore_weights reads the marker instead of guessing."""
    clean = {k: v for k, v in params.items() if isinstance(v, (int, float, str, bool)) or v is None}
    enc = params['encoder']
    verified = enc in t7.VERIFIED_ORDER
    entries = model._entries() if hasattr(model, '_entries') else model.fp.spec.entries
    named = model.get_parameters_dict()
    obj = {'modelW': t7.named_to_flat(named, entries, enc if verified else None), 'modelParams': clean}
    if optims:
        obj['optims'] = {'learningRate': float(model.optims['learningRate'])}
    if not verified:
        obj['vdLayout'] = 'declaration'
    t7.save(path, obj)
