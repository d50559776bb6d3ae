"""Decoder plug-ins, resolved by file name like model.lua:22-26.  A decoder module exposes
`declare(params, spec)`, `model(params, enc, fp, ws, drop)`, `forwardConnect`, `backwardConnect`
and optionally `decoderConnect` (model.lua:28-29)."""
import importlib

NAMES = ['disc', 'gen']


def load(name):
    if name not in NAMES:
        raise ValueError("unknown decoder '%s' (known: disc, gen)" % name)
    try:
        return importlib.import_module('visdial_amd.decoders.' + name)
    except ImportError as e:
        raise NotImplementedError("decoder '%s' is not built yet: %s" % (name, e))
