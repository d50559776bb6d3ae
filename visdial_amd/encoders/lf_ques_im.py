"""Counterpart of encoders/lf-ques-im.lua -- see _late_fusion.py (image part: True, history part: False)."""
from ._late_fusion import make

declare, model = make(use_im=True, use_hist=False)
