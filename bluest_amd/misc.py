"""
Host-side mirror of the hot-path part of bluest/misc.py (reference lines cited per function).  Same names,
argument meaning and error behaviour; every number comes from libbluest_hip.so (no CPU fallback).

The `*_c` functions reproduce the call shapes of the reference's native module `_cmisc_bluest`
(bluest/cmisc.cpp:99-110): they accumulate into their first argument in place and return None.
"""
import numpy as np

from . import _lib
from ._lib import check, ptr


def _c(a, dtype):
    return np.ascontiguousarray(a, dtype=dtype)


def _inplace(a, name):
    if not (isinstance(a, np.ndarray) and a.dtype == np.float64 and a.flags.c_contiguous and a.flags.writeable):
        # the reference silently copies such an argument and loses the result; be loud instead
        raise TypeError("%s must be a writeable C-contiguous float64 array (it is accumulated in place)" % name)
    return a


# ---- _cmisc_bluest call shapes ---------------------------------------------------------------------------

def assemble_psi_c(psi, N, k, Lk, groupsk, invcovsk):
    """cmisc.cpp:10-23"""
    g, ic = _c(groupsk, np.int64), _c(invcovsk, np.float64)
    check(_lib.lib().bluest_assemble_psi(ptr(_inplace(psi, "psi")), int(N), int(k), int(Lk), ptr(g), ptr(ic)))


def objectiveK_c(PHI, N, k, Lk, mk, groupsk, invcovsk):
    """cmisc.cpp:25-40; mk float64 or int64 (overloads :104-105)"""
    g, ic = _c(groupsk, np.int64), _c(invcovsk, np.float64)
    mk = np.ascontiguousarray(mk)
    if np.issubdtype(mk.dtype, np.integer):
        mk = _c(mk, np.int64)
        check(_lib.lib().bluest_objectiveK_i64(ptr(_inplace(PHI, "PHI")), int(N), int(k), int(Lk), ptr(mk), ptr(g), ptr(ic)))
    else:
        mk = _c(mk, np.float64)
        check(_lib.lib().bluest_objectiveK_f64(ptr(_inplace(PHI, "PHI")), int(N), int(k), int(Lk), ptr(mk), ptr(g), ptr(ic)))


def gradK_c(grad, k, Lk, groupsk, invcovsk, invPHI_0):
    """cmisc.cpp:58-72"""
    g, ic, v = _c(groupsk, np.int64), _c(invcovsk, np.float64), _c(invPHI_0, np.float64)
    check(_lib.lib().bluest_gradK(ptr(_inplace(grad, "grad")), int(k), int(Lk), ptr(g), ptr(ic), ptr(v), len(v)))


def cleanupK_c(X, k, Lk, groupsk, invcovsk, invPHI_0):
    """cmisc.cpp:42-56 (quirk of line 51 kept)"""
    g, ic, v = _c(groupsk, np.int64), _c(invcovsk, np.float64), _c(invPHI_0, np.float64)
    check(_lib.lib().bluest_cleanupK(ptr(_inplace(X, "X")), int(k), int(Lk), ptr(g), ptr(ic), ptr(v), len(v)))


def hessKQ_c(hess, N, k, q, Lk, Lq, groupsk, groupsq, invcovsk, invcovsq, invPHI):
    """cmisc.cpp:74-97"""
    gk, gq = _c(groupsk, np.int64), _c(groupsq, np.int64)
    ick, icq, P = _c(invcovsk, np.float64), _c(invcovsq, np.float64), _c(invPHI, np.float64)
    check(_lib.lib().bluest_hessKQ(ptr(_inplace(hess, "hess")), int(N), int(k), int(q), int(Lk), int(Lq), ptr(gk), ptr(gq),
                                   ptr(ick), ptr(icq), ptr(P)))


# ---- bluest/misc.py:600-629 wrappers ---------------------------------------------------------------------

def assemble_psi(N, k, Lk, groupsk, invcovsk):
    """misc.py:600-604"""
    psi = np.zeros((N * N, Lk), order="C")
    assemble_psi_c(psi.reshape(-1), N, k, Lk, np.asarray(groupsk).ravel(order="C"), invcovsk)
    return psi


def cleanupK(k, Lk, groupsk, invcovsk, invPHI):
    """misc.py:606-610"""
    N = invPHI.shape[0]
    X = np.zeros((N, Lk), order="C")
    cleanupK_c(X.reshape(-1), k, Lk, np.asarray(groupsk).ravel(order="C"), invcovsk, invPHI[0])
    return X


def objectiveK(N, k, Lk, mk, groupsk, invcovsk):
    """misc.py:612-616 as intended (the reference wrapper forgets N and raises TypeError)"""
    PHI = np.zeros((N * N,))
    objectiveK_c(PHI, N, k, Lk, mk, np.asarray(groupsk).ravel(order="C"), invcovsk)
    return PHI


def gradK(k, Lk, groupsk, invcovsk, invPHI):
    """misc.py:618-622"""
    grad = np.zeros((Lk,))
    gradK_c(grad, k, Lk, np.asarray(groupsk).ravel(order="C"), invcovsk, invPHI[0])
    return grad


def hessKQ(k, q, Lk, Lq, groupsk, groupsq, invcovsk, invcovsq, invPHI):
    """misc.py:624-629"""
    N = invPHI.shape[0]
    hess = np.zeros((Lk, Lq), order="C")
    hessKQ_c(hess.reshape(-1), N, k, q, Lk, Lq, np.asarray(groupsk).ravel(order="C"), np.asarray(groupsq).ravel(order="C"),
             invcovsk, invcovsq, np.asarray(invPHI).ravel(order="C"))
    return hess


def group_pinv(C, k, groupsk):
    """sap.py:69-79 for one group size: flat (Lk*k*k) pseudo-inverses of C[g,g], computed on the GPU"""
    C = _c(C, np.float64)
    g = _c(np.asarray(groupsk).ravel(order="C"), np.int64)
    Lk = len(g) // k
    out = np.empty(Lk * k * k, dtype=np.float64)
    check(_lib.lib().bluest_group_pinv(ptr(C), C.shape[0], int(k), int(Lk), ptr(g), ptr(out)))
    return out


def get_nnz_rows_cols(m, groups, cumsizes):
    """misc.py:453-457 (host index bookkeeping, used by PHIinvY0-style callers; the GPU path carries the same
    information as the per-model indicators of the Phi record)"""
    K = len(cumsizes) - 1
    ms = [m[cumsizes[k]:cumsizes[k + 1]] for k in range(K)]
    out = np.unique(np.concatenate([groups[k][abs(ms[k]) > 1.0e-6].flatten() for k in range(K)]))
    return out.reshape((len(out), 1)), out.reshape((1, len(out)))
