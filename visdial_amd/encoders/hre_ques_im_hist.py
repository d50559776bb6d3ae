"""Counterpart of encoders/hre-ques-im-hist.lua:5-97 -- see _hre.py (image part: True, history attention: False)."""
from ._hre import make

declare, model = make(use_im=True, attention=False)
