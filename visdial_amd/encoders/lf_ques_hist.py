"""Counterpart of encoders/lf-ques-hist.lua -- see _late_fusion.py (image part: False, history part: True)."""
from ._late_fusion import make

declare, model = make(use_im=False, use_hist=True)
