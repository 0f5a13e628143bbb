"""
bluest_amd -- MI355X-native implementation of BLUEST's sample-allocation hot path (Phi(m), V, grad V, SPG),
behind the reference's own operator API (SAP / MOSAP / BLUEProblem.setup_solver()/solve()).

Importing this package needs neither a GPU nor the compiled extension; the first compute call does, and it
raises loudly if either is missing (there is no CPU fallback).
"""
from ._lib import BluestHipError  # noqa: F401

__all__ = ["blue_fn", "SAP", "MOSAP", "BLUESTError", "BLUEProblem", "BluestHipError"]


def __getattr__(name):
    # lazy so that `import bluest_amd` (and bluest_amd.synth / .build) works without torch being imported
    if name == "blue_fn":
        from ._blue_fn import blue_fn
        return blue_fn
    if name == "SAP":
        from .sap import SAP
        return SAP
    if name in ("MOSAP",):
        from .mosap import MOSAP
        return MOSAP
    if name == "BLUESTError":
        from .sap import BLUESTError
        return BLUESTError
    if name == "BLUEProblem":
        from .blue_models import BLUEProblem
        return BLUEProblem
    raise AttributeError(name)
