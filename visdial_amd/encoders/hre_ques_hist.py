"""Counterpart of encoders/hre-ques-hist.lua -- see _hre.py (image part: False, history attention: False)."""
from ._hre import make

declare, model = make(use_im=False, attention=False)
