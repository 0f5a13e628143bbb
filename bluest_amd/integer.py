"""
Integer projection of a continuous allocation (bluest/sap.py:145-187, bluest/mosap.py:212-289,
bluest/misc.py:141-382): SURVEY.md section 8(f) row 1 -- the first "next" row after the hot path.
"""
from .sap import BLUESTError


def integer_projection_sap(sap, samples, budget=None, eps=None):
    raise BLUESTError("integer projection is not built yet in this round: call solve(..., continuous_relaxation=True)")


def integer_projection_mosap(mosap, samples, budget=None, eps=None):
    raise BLUESTError("integer projection is not built yet in this round: call solve(..., continuous_relaxation=True)")
