"""Counterpart of encoders/lf-ques-im-hist.lua:3-62 -- see _late_fusion.py (image part: True, history part: True)."""
from ._late_fusion import make

declare, model = make(use_im=True, use_hist=True)
