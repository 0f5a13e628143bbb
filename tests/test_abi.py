"""CPU tests: the C-ABI library builds, loads, and exports every symbol include/bluest_hip.h declares; compute
entry points fail loudly (BLUEST_ERR_NOGPU) instead of falling back when no GPU is present."""
import ctypes
import os
import re

import numpy as np
import pytest

from conftest import ROOT


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "bluest_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(bluest_[a-zA-Z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from bluest_amd import _lib, build
    build.build()
    L = ctypes.CDLL(_lib.LIB_PATH)
    syms = declared_symbols()
    assert len(syms) >= 25
    for s in syms:
        assert hasattr(L, s), "libbluest_hip.so does not export %s" % s
    # and the ctypes table binds exactly the declared set
    assert sorted(_lib.SIGNATURES) == syms
    assert _lib.lib().bluest_abi_version() == 1


def test_no_cpu_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from bluest_amd import _lib, misc
    assert _lib.device_count() == 0
    with pytest.raises(_lib.BluestHipError, match="no CPU path"):
        misc.objectiveK(3, 1, 2, np.ones(2), np.array([[0], [1]]), np.ones(2))
    with pytest.raises(_lib.BluestHipError):
        from bluest_amd.sap import SAP
        SAP(np.eye(3), 1, [[[0], [1], [2]]], np.ones(3), verbose=False)


def test_argument_errors_are_reported():
    from bluest_amd import _lib
    L = _lib.lib()
    h = ctypes.c_void_p()
    rc = L.bluest_plan_create(ctypes.byref(h), 1000, 10)
    assert rc == 1 and b"n_models" in L.bluest_last_error()
    assert L.bluest_plan_finalize(None, 1) == 1


def test_product_never_touches_the_oracle():
    """the oracle is test infrastructure: nothing under bluest_amd/ may import, load or mention it"""
    bad = []
    for dirpath, _, files in os.walk(os.path.join(ROOT, "bluest_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                if re.search(r"\boracle\b|liboracle|/root/reference", src):
                    bad.append(os.path.join(dirpath, f))
    assert not bad, bad
