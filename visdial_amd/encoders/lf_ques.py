"""Counterpart of encoders/lf-ques.lua:3-36 -- see _late_fusion.py (image part: False, history part: False)."""
from ._late_fusion import make

declare, model = make(use_im=False, use_hist=False)
