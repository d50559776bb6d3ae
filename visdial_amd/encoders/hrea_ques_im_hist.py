"""Counterpart of encoders/hrea-ques-im-hist.lua -- see _hre.py (image part: True, history attention: True)."""
from ._hre import make

declare, model = make(use_im=True, attention=True)
