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


def restore_weights(model, saved):
    w = saved['modelW']
    if saved.get('_flat_reference_layout'):
        model.load_flat_parameters(np.asarray(w, np.float32))
    else:
        model.wrapperW.copy_(w.to(model.wrapperW.device))


def save_t7(path, model, params):
    """write {modelW, optims, modelParams} in the reference's format (convert_gpu_to_cpu.lua's FloatTensor flavour)"""
    clean = {k: v for k, v in params.items() if isinstance(v, (int, float, str, bool)) or v is None}
    t7.save(path, {'modelW': model.flat_parameters(), 'optims': {'learningRate': float(model.optims['learningRate'])},
                   'modelParams': clean})
