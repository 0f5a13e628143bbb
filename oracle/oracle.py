"""
oracle/oracle.py -- numpy + plain-C restatement of the BLUEST sample-allocation hot path.

TEST INFRASTRUCTURE ONLY.  Importable only from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg; bluest_amd/ never imports it (tests/test_abi.py::test_product_never_touches_the_oracle enforces that).
Parity status: PINNED by tests/golden/*.npz, generated from the real reference by oracle/gen_golden.py.

Layers restated (paths relative to /root/reference/):
  L0  bluest/cmisc.cpp            -> liboracle_bluest.so (oracle/bluest_oracle.c), wrapped below
  L1  bluest/misc.py:453-495,600-629 -> get_nnz_rows_cols, get_phi_full, variance_full, variance_GH_full, ...
  L2  bluest/sap.py:52-143        -> OracleSAP
  L3  bluest/mosap.py:20-100      -> OracleMOSAP
      bluest/spg.py:3-132         -> linesearch, spg
Third-party arithmetic on the path is the same numpy.linalg the reference calls (pinv/solve); the C file
carries its own Jacobi/LU stand-ins so the pure-C baseline needs no LAPACK.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

_i64p = ctypes.POINTER(ctypes.c_int64)
_f64p = ctypes.POINTER(ctypes.c_double)
_u8p = ctypes.POINTER(ctypes.c_uint8)


def build(force=False):
    """Compile the C restatement (gcc; strict-IEEE build for the checker, fast-math twin for the timed CPU baseline) and, when
    the reference tree is present, oracle/_ref."""
    src = os.path.join(_HERE, "bluest_oracle.c")
    for name in ("liboracle_bluest_strict.so", "liboracle_bluest.so"):
        so = os.path.join(_HERE, name)
        if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
            subprocess.check_call(["make", "-C", _HERE, name], stdout=subprocess.DEVNULL)
    if os.path.exists("/root/reference/bluest/cmisc.cpp"):
        subprocess.check_call(["make", "-C", _HERE, "ref"], stdout=subprocess.DEVNULL)
    return os.path.join(_HERE, "liboracle_bluest_strict.so")


_FAST = False


def select_fast_math():
    """bench.py's cpu_baseline leg only: switch to the build with the reference's own compiler flags (setup.py:6, -ffast-math).
    Loading a -ffast-math shared object sets FTZ/DAZ for the whole process (crtfastmath), so the CHECKER never does: tests and
    smoke() stay on the strict-IEEE build."""
    global _FAST, _LIB
    if not _FAST:
        _FAST, _LIB = True, None


def select_strict():
    """back to the strict-IEEE build after a timed leg (the FTZ/DAZ bits a fast-math object set at load time stay set for the
    process; what this restores is WHICH object the wrappers call)"""
    global _FAST, _LIB
    if _FAST:
        _FAST, _LIB = False, None


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "liboracle_bluest.so" if _FAST else "liboracle_bluest_strict.so")
        if not os.path.exists(so):
            build()
        _LIB = ctypes.CDLL(so)
        _LIB.orc_sym_pinv.restype = ctypes.c_int
        _LIB.orc_solve.restype = ctypes.c_int
        _LIB.orc_variance.restype = ctypes.c_int
        _LIB.orc_variance_GH.restype = ctypes.c_int
    return _LIB


_REF = None


def ref_native():
    """the reference's OWN native module, compiled from /root/reference/bluest/cmisc.cpp into oracle/_ref/ by
    `make -C oracle ref` (build container only; the binary travels to the GPU box, the source does not).  None if absent.
    Built with the reference's flags (-ffast-math): importing it sets FTZ/DAZ for the process, so only bench.py's cpu_baseline,
    gen_golden.py and a SUBPROCESS of the test suite load it."""
    global _REF
    if _REF is None:
        import glob
        import importlib.util
        cands = glob.glob(os.path.join(_HERE, "_ref", "_cmisc_bluest*.so"))
        if not cands:
            return None
        spec = importlib.util.spec_from_file_location("_cmisc_bluest", cands[0])
        _REF = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(_REF)
    return _REF


def _f(a):
    a = np.ascontiguousarray(a, dtype=np.float64)
    return a, a.ctypes.data_as(_f64p)


def _i(a):
    a = np.ascontiguousarray(a, dtype=np.int64)
    return a, a.ctypes.data_as(_i64p)


# --------------------------------------------------------------------------------------------------
# L0 wrappers: same call shapes as bluest/misc.py:600-629
# --------------------------------------------------------------------------------------------------

def assemble_psi(N, k, Lk, groupsk, invcovsk):
    """bluest/misc.py:600-604 -> cmisc.cpp:10-23"""
    psi = np.zeros((N * N, Lk), order="C")
    g, gp = _i(np.asarray(groupsk).ravel(order="C"))
    ic, icp = _f(invcovsk)
    lib().orc_assemble_psi(psi.ctypes.data_as(_f64p), int(N), int(k), ctypes.c_int64(Lk), gp, icp)
    return psi


def objectiveK(N, k, Lk, mk, groupsk, invcovsk):
    """what bluest/misc.py:612-616 intends (the reference wrapper omits N and raises TypeError); cmisc.cpp:25-40"""
    PHI = np.zeros((N * N,))
    g, gp = _i(np.asarray(groupsk).ravel(order="C"))
    ic, icp = _f(invcovsk)
    mk = np.ascontiguousarray(mk)
    if np.issubdtype(mk.dtype, np.integer):
        mm, mp = _i(mk)
        lib().orc_objectiveK_i64(PHI.ctypes.data_as(_f64p), int(N), int(k), ctypes.c_int64(Lk), mp, gp, icp)
    else:
        mm, mp = _f(mk)
        lib().orc_objectiveK_f64(PHI.ctypes.data_as(_f64p), int(N), int(k), ctypes.c_int64(Lk), mp, gp, icp)
    return PHI


def gradK(k, Lk, groupsk, invcovsk, invPHI):
    """bluest/misc.py:618-622 -> cmisc.cpp:58-72 (uses invPHI[0])"""
    grad = np.zeros((Lk,))
    g, gp = _i(np.asarray(groupsk).ravel(order="C"))
    ic, icp = _f(invcovsk)
    v, vp = _f(invPHI[0])
    lib().orc_gradK(grad.ctypes.data_as(_f64p), int(k), ctypes.c_int64(Lk), gp, icp, vp)
    return grad


def cleanupK(k, Lk, groupsk, invcovsk, invPHI):
    """bluest/misc.py:606-610 -> cmisc.cpp:42-56 (with the `=` quirk of line 51)"""
    N = invPHI.shape[0]
    X = np.zeros((N, Lk), order="C")
    g, gp = _i(np.asarray(groupsk).ravel(order="C"))
    ic, icp = _f(invcovsk)
    v, vp = _f(invPHI[0])
    lib().orc_cleanupK(X.ctypes.data_as(_f64p), int(k), ctypes.c_int64(Lk), gp, icp, vp)
    return X


def hessKQ(k, q, Lk, Lq, groupsk, groupsq, invcovsk, invcovsq, invPHI):
    """bluest/misc.py:624-629 -> cmisc.cpp:74-97"""
    N = invPHI.shape[0]
    hess = np.zeros((Lk, Lq), order="C")
    gk, gkp = _i(np.asarray(groupsk).ravel(order="C"))
    gq, gqp = _i(np.asarray(groupsq).ravel(order="C"))
    ick, ickp = _f(invcovsk)
    icq, icqp = _f(invcovsq)
    P, Pp = _f(np.asarray(invPHI).ravel(order="C"))
    lib().orc_hessKQ(hess.ctypes.data_as(_f64p), int(N), int(k), int(q), ctypes.c_int64(Lk), ctypes.c_int64(Lq),
                     gkp, gqp, ickp, icqp, Pp)
    return hess


# --------------------------------------------------------------------------------------------------
# L1: bluest/misc.py:453-495
# --------------------------------------------------------------------------------------------------

def get_nnz_rows_cols(m, groups, cumsizes):
    """bluest/misc.py:453-457"""
    K = len(cumsizes) - 1
    ms = [m[cumsizes[k]:cumsizes[k + 1]] for k in range(K)]
    out = np.unique(np.concatenate([groups[k][abs(ms[k]) > 1.0e-6].flatten() for k in range(K)]))
    return out.reshape((len(out), 1)), out.reshape((1, len(out)))


def get_phi_full(m, psi, delta=0.0):
    """bluest/misc.py:459-461"""
    N = int(round(np.sqrt(psi.shape[0])))
    return delta * np.eye(N) + (psi @ m).reshape((N, N))


def variance_full(m, psi, groups, cumsizes, delta=0.0):
    """bluest/misc.py:463-477"""
    if abs(m).max() < 0.05:
        return np.inf
    PHI = get_phi_full(m, psi, delta=delta)
    idx = get_nnz_rows_cols(m, groups, cumsizes)
    PHI = PHI[idx]
    assert idx[0].min() == 0  # misc.py:470
    return np.linalg.solve(PHI, np.eye(len(idx[0]), 1).flatten())[0]


def variance_GH_full(m, psi, groups, sizes, invcovs, delta=0.0, nohess=False):
    """bluest/misc.py:479-505"""
    K = len(groups)
    L = len(m)
    cumsizes = np.cumsum(sizes)
    if abs(m).max() < 0.05:
        return np.inf, np.inf * np.ones((L,))
    PHI = get_phi_full(m, psi, delta=delta)
    invPHI = np.linalg.pinv(PHI)
    idx = get_nnz_rows_cols(m, groups, cumsizes)
    var = np.linalg.pinv(PHI[idx])[0, 0]
    grad = -np.concatenate([gradK(k, sizes[k], groups[k - 1], invcovs[k - 1], invPHI) for k in range(1, K + 1)])
    if nohess:
        return var, grad, None
    hess = np.zeros((L, L))
    for k in range(1, K + 1):
        for q in range(1, K + 1):
            hess[cumsizes[k - 1]:cumsizes[k], :][:, cumsizes[q - 1]:cumsizes[q]] = hessKQ(
                k, q, sizes[k], sizes[q], groups[k - 1], groups[q - 1], invcovs[k - 1], invcovs[q - 1], invPHI)
    hess += hess.T
    return var, grad, hess


# --------------------------------------------------------------------------------------------------
# L2: bluest/sap.py:52-143
# --------------------------------------------------------------------------------------------------

class OracleSAP(object):
    """bluest/sap.py:52-143 (constructor + get_variance_functions), numpy.linalg.pinv per group as at :74."""

    def __init__(self, C, K, groups, costs):
        self.C = C
        self.N = C.shape[0]
        self.K = K
        self.costs = costs
        invcovs = [[] for k in range(K)]
        sizes = [0] + [len(groupsk) for groupsk in groups]
        groups = list(groups)
        for k in range(1, K + 1):
            gk = np.array(groups[k - 1], dtype=np.int64).reshape((-1, k))
            groups[k - 1] = gk
            if len(gk) > 0:
                sub = C[gk[:, :, None], gk[:, None, :]]          # (Lk,k,k) = C[idx.T, idx] per group (sap.py:72-74)
                invcovs[k - 1] = np.concatenate([np.linalg.pinv(sub[i]).ravel() for i in range(len(gk))])
            else:
                invcovs[k - 1] = np.array([])
        self.sizes = sizes
        self.groups = groups
        self.invcovs = invcovs
        self.cumsizes = np.cumsum(sizes)
        self.L = int(self.cumsizes[-1])
        allg = [g for gk in groups for g in gk]
        self.ES = [np.array([int(i in g) for g in allg]) for i in range(self.N)]   # sap.py:89-94
        self.e = self.ES[0]
        self.psi = np.hstack([assemble_psi(self.N, k, sizes[k], groups[k - 1], invcovs[k - 1])
                              for k in range(1, K + 1) if len(groups[k - 1]) > 0])  # sap.py:129

    def get_phi(self, m, delta=0):
        return get_phi_full(m, self.psi, delta=delta)

    def variance(self, m, delta=0):
        return variance_full(m, self.psi, self.groups, self.cumsizes, delta=delta)

    def variance_GH(self, m, delta=0, nohess=False):
        return variance_GH_full(m, self.psi, self.groups, self.sizes, self.invcovs, delta=delta, nohess=nohess)

    def compute_BLUE_estimator(self, sums, samples):
        """bluest/sap.py:99-119 (y = sum_i R_i^T C_i^-1 sums_i, looped over ALL groups as the reference does) followed by
        bluest/misc.py:518-544 (`PHIinvY0`: mu = sum_j pinv(Phi[idx])[0, j] y_j, var = pinv(Phi[idx])[0, 0])"""
        K, sizes, cumsizes, groups, invcovs = self.K, self.sizes, self.cumsizes, self.groups, self.invcovs
        y = [0 for i in range(self.L)]                     # the reference allocates L entries and uses the first N (sap.py:111)
        per_size = [sums[cumsizes[k]:cumsizes[k + 1]] for k in range(K)]
        for k in range(1, K + 1):
            for i in range(sizes[k]):
                for j in range(k):
                    for t in range(k):
                        y[groups[k - 1][i][j]] += invcovs[k - 1][k * k * i + k * j + t] * per_size[k - 1][i][t]
        m = np.asarray(samples)
        if abs(m).max() < 0.05:
            return np.inf
        PHI = get_phi_full(m, self.psi)
        idx = get_nnz_rows_cols(m, groups, cumsizes)
        PHI = PHI[idx]
        yy = [y[item] for item in idx[0].flatten()]
        assert idx[0].min() == 0
        pinvPHI = np.linalg.pinv(PHI)
        mu = 0
        for j in range(len(yy)):
            mu += pinvPHI[0, j] * yy[j]
        return mu, pinvPHI[0, 0]

    def variance_GH_as_executed(self, m, delta=0.0):
        """bluest/misc.py:479-495 (nohess) exactly as the reference executes it on a CPU: dense psi@m (BLAS dgemv), two
        numpy pinv calls, and the reference's own compiled gradK_c (oracle/_ref).  Used as bench.py's cpu_baseline of kind
        "reference".  Raises if oracle/_ref is not there."""
        cm = ref_native()
        if cm is None:
            raise RuntimeError("oracle/_ref is absent")
        K, sizes, groups, invcovs = self.K, self.sizes, self.groups, self.invcovs
        PHI = get_phi_full(m, self.psi, delta=delta)
        invPHI = np.linalg.pinv(PHI)
        idx = get_nnz_rows_cols(m, groups, self.cumsizes)
        var = np.linalg.pinv(PHI[idx])[0, 0]
        out = []
        for k in range(1, K + 1):
            grad = np.zeros((sizes[k],))
            cm.gradK_c(grad, k, sizes[k], groups[k - 1].ravel(order='C'), invcovs[k - 1], invPHI[0])
            out.append(grad)
        return var, -np.concatenate(out), None

    # ---- flat views for the pure-C twins -------------------------------------------------------
    def flat(self):
        sizes = np.asarray(self.sizes, dtype=np.int64)
        groups = np.concatenate([g.ravel() for g in self.groups]).astype(np.int64)
        invcovs = np.concatenate([np.asarray(ic, dtype=np.float64).ravel() for ic in self.invcovs])
        return sizes, groups, invcovs

    def c_variance(self, m, delta=0.0, dense=False):
        """pure-C twin of variance (LU solve); dense=True uses the dense psi GEMV as the reference executes"""
        sizes, groups, invcovs = self.flat()
        m, mp = _f(m)
        var = ctypes.c_double(0.0)
        psi_p = self.psi.ctypes.data_as(_f64p) if dense else None
        rc = lib().orc_variance(mp, psi_p, int(self.N), int(self.K), sizes.ctypes.data_as(_i64p),
                                groups.ctypes.data_as(_i64p), invcovs.ctypes.data_as(_f64p),
                                ctypes.c_double(delta), ctypes.byref(var))
        return rc, var.value

    def c_variance_GH(self, m, delta=0.0, dense=False):
        """pure-C twin of variance_GH(nohess=True) (Jacobi pinv)"""
        sizes, groups, invcovs = self.flat()
        m, mp = _f(m)
        var = ctypes.c_double(0.0)
        grad = np.zeros(self.L)
        v = np.zeros(self.N)
        PHI = np.zeros((self.N, self.N))
        psi_p = self.psi.ctypes.data_as(_f64p) if dense else None
        rc = lib().orc_variance_GH(mp, psi_p, int(self.N), int(self.K), sizes.ctypes.data_as(_i64p),
                                   groups.ctypes.data_as(_i64p), invcovs.ctypes.data_as(_f64p),
                                   ctypes.c_double(delta), ctypes.byref(var), grad.ctypes.data_as(_f64p),
                                   v.ctypes.data_as(_f64p), PHI.ctypes.data_as(_f64p))
        return rc, var.value, grad, v, PHI


def c_group_pinv(C, k, groupsk):
    """pure-C twin of sap.py:69-79 for one group size (Jacobi pinv)"""
    C, Cp = _f(C)
    g, gp = _i(np.asarray(groupsk).ravel())
    Lk = len(g) // k
    out = np.zeros(Lk * k * k)
    lib().orc_group_pinv(Cp, int(C.shape[0]), int(k), ctypes.c_int64(Lk), gp, out.ctypes.data_as(_f64p))
    return out


class SparseOracleSAP(object):
    """OracleSAP without the dense psi (n^2 x K_tot: 1.2 GB at n = 25, K_tot = 245505): Phi comes from the sparse loop
    `objectiveK_c` (bluest/cmisc.cpp:25-40), which sums the same products psi@m does; everything after Phi is
    bluest/misc.py:463-495 verbatim in behaviour (numpy solve / pinv on the sampled models, gradK_c).  The per-group
    pseudo-inverses are numpy.linalg.pinv on the stacked (L_k, k, k) blocks (bluest/sap.py:69-79, one LAPACK call per size).
    Used to judge allocations at sizes where OracleSAP is too heavy for a test."""

    def __init__(self, C, K, groups):
        self.C, self.N, self.K = np.asarray(C, dtype=np.float64), C.shape[0], K
        self.groups = [np.asarray(g, dtype=np.int64).reshape((-1, k + 1)) for k, g in enumerate(groups)]
        self.sizes = [0] + [len(g) for g in self.groups]
        self.cumsizes = np.cumsum(self.sizes)
        self.L = int(self.cumsizes[-1])
        self.invcovs = []
        for gk in self.groups:
            if len(gk):
                sub = self.C[gk[:, :, None], gk[:, None, :]]
                self.invcovs.append(np.linalg.pinv(sub).reshape(-1))
            else:
                self.invcovs.append(np.zeros(0))

    def get_phi(self, m, delta=0.0):
        N = self.N
        PHI = np.zeros(N * N)
        for k in range(1, self.K + 1):
            if self.sizes[k]:
                PHI += objectiveK(N, k, self.sizes[k], m[self.cumsizes[k - 1]:self.cumsizes[k]], self.groups[k - 1], self.invcovs[k - 1])
        return delta * np.eye(N) + PHI.reshape(N, N)

    def variance(self, m, delta=0.0):
        """bluest/misc.py:463-477"""
        if abs(m).max() < 0.05:
            return np.inf
        PHI = self.get_phi(m, delta)
        idx = get_nnz_rows_cols(m, self.groups, self.cumsizes)
        assert idx[0].min() == 0
        return np.linalg.solve(PHI[idx], np.eye(len(idx[0]), 1).flatten())[0]

    def variance_GH(self, m, delta=0.0, nohess=True):
        """bluest/misc.py:479-495 (nohess)"""
        if abs(m).max() < 0.05:
            return np.inf, np.inf * np.ones(self.L)
        PHI = self.get_phi(m, delta)
        invPHI = np.linalg.pinv(PHI)
        idx = get_nnz_rows_cols(m, self.groups, self.cumsizes)
        var = np.linalg.pinv(PHI[idx])[0, 0]
        grad = -np.concatenate([gradK(k, self.sizes[k], self.groups[k - 1], self.invcovs[k - 1], invPHI)
                                for k in range(1, self.K + 1) if self.sizes[k]])
        return var, grad, None


def multiplier_certificate(saps, m, costs, mu, s=None, eps_in=1.0e-3):
    """Weak-duality bound for  min F(m') = max_o V_o(m')/s_o  s.t.  costs.m' = B = costs.m, m' >= 0  from GIVEN multipliers mu (any
    point of the simplex gives a valid bound) and y_o = row 0 of the inverse information matrix at the slightly interior point
    (1 - eps_in) m + eps_in * (B / costs / K_tot)  (every model sampled there, so y has no blind components):
        F* >= LB = A^2 / (4 B max_i c_i),   A = 2 sum_o (mu_o/s_o) y_{o,0},   c_i = sum_o (mu_o/s_o) y_{o,g_i}^T C_{i,o}^-1 y_{o,g_i} / w_i
    (see optimality_certificate).  Everything is evaluated here: Phi by the objectiveK_c loop, its inverse by numpy, the
    quadratic forms of ALL groups by gradK_c.  Returns (relative gap (F - LB)/F, F, LB)."""
    O = len(saps)
    m = np.asarray(m, dtype=np.float64)
    w = np.asarray(costs, dtype=np.float64)
    s = np.ones(O) if s is None else np.asarray(s, dtype=np.float64)
    mu = np.maximum(np.asarray(mu, dtype=np.float64), 0.0)
    mu = mu / mu.sum()
    B = float(w @ m)
    F = max(q.variance(m) / so for q, so in zip(saps, s))
    mi = (1.0 - eps_in) * m + eps_in * (B / w / len(m))
    a = mu / s
    A, ci = 0.0, np.zeros(len(m))
    for o, q in enumerate(saps):
        if a[o] == 0.0:
            continue
        # y = Phi(mi)^-1 e_0 by a diagonally scaled solve (NOT numpy's pinv: its relative cut-off drops the directions that only
        # the tiny interior weight samples, which is exactly where y must not be blind); any y gives a valid bound
        PHI = q.get_phi(mi)
        dsc = 1.0 / np.sqrt(np.diag(PHI))
        y = dsc * np.linalg.solve(PHI * np.outer(dsc, dsc), dsc * np.eye(q.N, 1).ravel())
        quad = np.concatenate([gradK(k, q.sizes[k], q.groups[k - 1], q.invcovs[k - 1], y[None, :])
                               for k in range(1, q.K + 1) if q.sizes[k]])
        A += 2.0 * a[o] * y[0]
        ci += a[o] * quad / w
    lb = A * A / (4.0 * B * float(ci.max()))
    return (F - lb) / F, F, lb


def multiplier_certificate_ragged(saps, maps, m, costs, mu, s=None, eps_in=1.0e-3):
    """multiplier_certificate for outputs with their OWN group lists (bluest/mosap.py:54-67): saps[o] is built on output o's groups,
    maps[o][j] is the position of its j-th group in the global list that m and costs refer to.  Same bound: a group that output o
    does not use contributes nothing to c_i for that output.  Returns (relative gap, F, LB)."""
    O = len(saps)
    m = np.asarray(m, dtype=np.float64)
    w = np.asarray(costs, dtype=np.float64)
    s = np.ones(O) if s is None else np.asarray(s, dtype=np.float64)
    mu = np.maximum(np.asarray(mu, dtype=np.float64), 0.0)
    mu = mu / mu.sum()
    B = float(w @ m)
    F = max(q.variance(m[mp]) / so for q, mp, so in zip(saps, maps, s))
    mi = (1.0 - eps_in) * m + eps_in * (B / w / len(m))
    a = mu / s
    A, ci = 0.0, np.zeros(len(m))
    for o, (q, mp) in enumerate(zip(saps, maps)):
        if a[o] == 0.0:
            continue
        PHI = q.get_phi(mi[mp])
        dsc = 1.0 / np.sqrt(np.diag(PHI))
        y = dsc * np.linalg.solve(PHI * np.outer(dsc, dsc), dsc * np.eye(q.N, 1).ravel())
        quad = np.concatenate([gradK(k, q.sizes[k], q.groups[k - 1], q.invcovs[k - 1], y[None, :])
                               for k in range(1, q.K + 1) if q.sizes[k]])
        A += 2.0 * a[o] * y[0]
        ci[mp] += a[o] * quad / w[mp]
    lb = A * A / (4.0 * B * float(ci.max()))
    return (F - lb) / F, F, lb


def optimality_certificate(saps, m, costs, s=None, tol=1.0e-7, max_rounds=60, verbose=False, max_seconds=None):
    """Duality certificate for  min_m F(m) = max_o V_o(m)/s_o  s.t.  costs.m = B, m >= 0  (the problem bluest/sap.py:387-418
    and bluest/mosap.py:578-605 hand to scipy) at a candidate allocation m, B = costs.m.  saps: one SparseOracleSAP per
    output, all on the same global group list.

    Weak duality.  V_o(m') = sup_y 2 y_0 - sum_i m'_i q_{i,o}(y) with q_{i,o}(y) = y_g^T C_{g,o}^-1 y_g >= 0 (Phi_o is linear in
    m').  Hence for ANY vectors y_o in R^n and ANY mu in the simplex, with a_o = mu_o / s_o, every feasible m' has
        F(m') >= sum_o mu_o V_o(m')/s_o >= A - B max_i c_i,    A = sum_o a_o 2 y_{o,0},   c_i = sum_o a_o q_{i,o}(y_o) / w_i,
    and scaling all y_o by the best common factor gives  LB = A^2 / (4 B max_i c_i).  The bound is VALID whatever (mu, y) are
    and TIGHT at the dual optimum (the problem is convex, Slater holds).  A first-order choice -- y from the gradient at m -- is
    useless in practice: V is so strongly curved in the cheap groups that a point 1e-4 above the optimum still has Lagrange
    multipliers off by a factor 2-3.  So the dual is SOLVED here: for fixed mu, minimise t subject to c_i(y) <= t for all groups
    at fixed A (a convex QCQP in n_active * n variables; constraints generated on demand, SLSQP on the working set, started from
    row 0 of the inverse of Phi_o(m) on the sampled models = the reference's v, bluest/misc.py:487); mu, which lives on the outputs
    within 2 % of the maximum, is optimised by a bounded scalar search when two outputs are in play (the dual function is
    concave in mu) and by cyclic pairwise searches beyond.  Only quantities the oracle computes enter (Phi via objectiveK_c, the
    quadratic forms via gradK_c); nothing from the GPU path.  Returns (relative gap (F - LB)/F, LB, mu, info)."""
    import time
    from scipy.optimize import minimize, minimize_scalar
    deadline = None if max_seconds is None else time.time() + float(max_seconds)
    O = len(saps)
    m = np.asarray(m, dtype=np.float64)
    w = np.asarray(costs, dtype=np.float64)
    s = np.ones(O) if s is None else np.asarray(s, dtype=np.float64)
    B = float(w @ m)
    sap0 = saps[0]
    N, K, L = sap0.N, sap0.K, sap0.L
    Vs = np.array([q.variance(m) for q in saps])
    r = Vs / s
    F = float(r.max())
    act = np.flatnonzero(r >= 0.98 * F)
    nA = len(act)
    sa = s[act]
    Y0 = np.zeros((nA, N))                     # y_o / F: normalised so that y_{o,0} = V_o / F = O(1)
    for c, o in enumerate(act):
        PHI = saps[o].get_phi(m)
        d = np.diag(PHI)
        S = np.flatnonzero(d > 1.0e-13 * d.max())
        assert S[0] == 0, "model 0 is not sampled"
        Y0[c, S] = np.linalg.solve(PHI[np.ix_(S, S)], np.eye(len(S), 1).ravel()) / F

    def quad_forms(Yc):
        """B F q_{i,o}(y_o) / w_i for all groups i and the active outputs: (L, nA); cmisc.cpp:58-72 evaluates this form"""
        res = np.empty((L, nA))
        for c, o in enumerate(act):
            q = saps[o]
            res[:, c] = np.concatenate([gradK(k, q.sizes[k], q.groups[k - 1], q.invcovs[k - 1], Yc[c][None, :])
                                        for k in range(1, K + 1) if q.sizes[k]])
        return res * (B * F) / w[:, None]

    blocks = {}
    kmax = max(k for k in range(1, K + 1) if sap0.sizes[k])

    def sparse_blocks(idx):
        """for the groups i = idx[j]: their models (padded with N, which addresses a zero appended to y) and, per active
        output, the k x k block B F C_{i,o}^-1 / w_i (zero-padded to kmax x kmax)"""
        gi = np.full((len(idx), kmax), N, dtype=np.int64)
        bl = np.zeros((len(idx), nA, kmax, kmax))
        for j, i in enumerate(idx):
            if i not in blocks:
                k = int(np.searchsorted(sap0.cumsizes, i, side="right"))
                li = i - sap0.cumsizes[k - 1]
                g = np.full(kmax, N, dtype=np.int64)
                g[:k] = sap0.groups[k - 1][li]
                b = np.zeros((nA, kmax, kmax))
                for c, o in enumerate(act):
                    b[c, :k, :k] = saps[o].invcovs[k - 1][li * k * k:(li + 1) * k * k].reshape(k, k) * (B * F / w[i])
                blocks[i] = (g, b)
            gi[j], bl[j] = blocks[i]
        return gi, bl

    nY = nA * N
    state = {"work": np.zeros(0, dtype=np.int64), "Y": Y0.copy(), "solves": 0, "best": (-np.inf, None, None)}

    def dual_for(mu_a):
        """max over y of the normalised bound LB/F for these multipliers (warm-started from the previous call)"""
        a = mu_a / sa
        a_grad = np.zeros(nY + 1)
        a_grad[np.arange(nA) * N] = 2.0 * a
        e_tau = np.zeros(nY + 1)
        e_tau[-1] = 1.0
        Yc = state["Y"].copy()
        A0 = float((2.0 * a * Yc[:, 0]).sum())
        work = state["work"]
        best_lb, solved = -np.inf, False
        tau = 0.0
        for rnd in range(max_rounds):
            c = quad_forms(Yc) @ a
            A = float((2.0 * a * Yc[:, 0]).sum())
            lb = A * A / (4.0 * c.max())
            if lb > best_lb:
                best_lb = lb
                if lb > state["best"][0]:
                    state["best"] = (lb, Yc.copy(), mu_a.copy())
                state["Y"] = Yc.copy()
            viol = np.setdiff1d(np.flatnonzero(c > tau * (1.0 + tol) + 1.0e-14), work)
            if len(viol) == 0 and solved:
                break
            add = viol[np.argsort(-c[viol])[:96]] if len(viol) else np.zeros(0, dtype=np.int64)
            if len(work) > 600:                                      # keep the QP small: constraints far from active leave the working
                work = work[c[work] >= 0.8 * max(tau, 1.0e-300)]     # set (they come back through `viol` if they matter again)
            work = np.unique(np.concatenate([work, add]))
            if deadline is not None and time.time() > deadline:      # out of time: the best bound so far stays valid
                break
            gi, bl = sparse_blocks(work)
            rows_j = np.arange(len(work))

            def cons(xv, gi=gi, bl=bl):
                Yv = np.concatenate([xv[:nY].reshape(nA, N), np.zeros((nA, 1))], axis=1)
                Yg = Yv[:, gi]                                                     # (nA, j, kmax)
                t = np.einsum("jckl,cjl->cjk", bl, Yg)
                return xv[-1] - np.einsum("c,cjk,cjk->j", a, t, Yg)

            def cons_jac(xv, gi=gi, bl=bl, rows_j=rows_j):
                Yv = np.concatenate([xv[:nY].reshape(nA, N), np.zeros((nA, 1))], axis=1)
                Yg = Yv[:, gi]
                t = np.einsum("jckl,cjl->cjk", bl, Yg)                              # (nA, j, kmax): C^-1 y_g
                J3 = np.zeros((len(gi), nA, N + 1))
                for kk in range(kmax):
                    np.add.at(J3, (rows_j[:, None], np.arange(nA)[None, :], gi[:, kk][:, None]), (-2.0 * a[:, None] * t[:, :, kk]).T)
                J = np.empty((len(gi), nY + 1))
                J[:, :nY] = J3[:, :, :N].reshape(len(gi), nY)
                J[:, -1] = 1.0
                return J

            x0 = np.concatenate([Yc.ravel(), [float(c[work].max())]])
            res = minimize(lambda xv: xv[-1], x0, jac=lambda xv: e_tau, method="SLSQP",
                           constraints=[{"type": "ineq", "fun": cons, "jac": cons_jac},
                                        {"type": "eq", "fun": lambda xv: float(a_grad @ xv) - A0, "jac": lambda xv: a_grad}],
                           options={"maxiter": 300, "ftol": 1.0e-15})
            state["solves"] += 1
            # SLSQP often stops with "positive directional derivative" (status 8) AT the solution of such min-max problems:
            # what counts is that the point is feasible for the working set and not worse than the start
            ok = np.isfinite(res.x).all() and res.x[-1] <= x0[-1] * (1.0 + 1.0e-9) and \
                cons(res.x).min() >= -1.0e-6 * max(res.x[-1], 1.0e-300) and abs(float(a_grad @ res.x) - A0) <= 1.0e-8 * abs(A0)
            solved = ok and res.status in (0, 8)
            if ok:
                Yc, tau = res.x[:nY].reshape(nA, N), float(res.x[-1])
            if verbose:
                print("   mu %s round %d: working set %d, SLSQP status %d nit %d, tau %.10f, bound %.10f"
                      % (np.round(mu_a, 5), rnd, len(work), res.status, res.nit, tau, best_lb))
        state["work"] = work
        return best_lb

    if nA == 1:
        dual_for(np.ones(1))
    else:
        # multipliers from the stationarity of the candidate on its own support: sum_o mu_o grad r_o(m)_i / w_i = -lambda for the
        # groups with m_i > 0 (there the pinv-based gradient is exact: those groups only contain sampled models) -- a small
        # non-negative least-squares problem; exact at an optimum, a good start near one
        from scipy.optimize import nnls
        sup = np.flatnonzero(m > 1.0e-9 * m.max())
        Gs = quad_forms(Y0)[sup] / (B * F) * (F * F)             # q_{i,o}(y_o) / w_i with y = F * Y0, i.e. -grad V_o / w_i
        Gs = Gs / sa[None, :]
        scale_rows = 1.0 / np.abs(Gs).max()
        A_ls = np.vstack([np.hstack([Gs * scale_rows, -np.ones((len(sup), 1))]), np.concatenate([np.ones(nA), [0.0]])[None, :] * 10.0])
        b_ls = np.concatenate([np.zeros(len(sup)), [10.0]])
        sol, _ = nnls(A_ls, b_ls)
        mu_a = np.maximum(sol[:nA], 1.0e-9)
        mu_a /= mu_a.sum()
        dual_for(mu_a.copy())
        for sweep in range(2):
            for c in range(1, nA):                                   # move weight between output act[0] and act[c]
                if deadline is not None and time.time() > deadline:
                    break
                tot = mu_a[0] + mu_a[c]
                t0 = mu_a[c] / tot

                def along(t, c=c, tot=tot):
                    mm = mu_a.copy()
                    mm[0], mm[c] = tot * (1.0 - t), tot * t
                    return -dual_for(np.maximum(mm, 1.0e-12))

                res = minimize_scalar(along, bounds=(max(0.0, t0 - 0.15), min(1.0, t0 + 0.15)), method="bounded",
                                      options={"xatol": 2.0e-4, "maxiter": 10})
                mu_a[0], mu_a[c] = tot * (1.0 - res.x), tot * res.x
            if state["best"][0] > 1.0 - 1.0e-5:
                break
    lb_n, Yb, mub = state["best"]
    lb = lb_n * F
    mu_full = np.zeros(O)
    mu_full[act] = mub
    info = {"active_outputs": act.tolist(), "working_set": int(len(state["work"])), "qp_solves": state["solves"]}
    if verbose:
        print("certificate: F = %.10e, LB = %.10e, gap = %.3e, mu = %s, %d QP solves, working set %d"
              % (F, lb, (F - lb) / F, np.round(mu_full, 5), state["solves"], len(state["work"])))
    return (F - lb) / F, lb, mu_full, info


# --------------------------------------------------------------------------------------------------
# L3: bluest/mosap.py:20-100 (hash-based mapping instead of the O(L_k^2) search at :54-65; same result)
# --------------------------------------------------------------------------------------------------

class OracleMOSAP(object):
    def __init__(self, C, K, Ks, groups, multi_groups, costs, multi_costs):
        self.n_outputs = len(C)
        self.N = C[0].shape[0]
        self.K = K
        self.Ks = Ks
        self.costs = costs
        self.groups = [np.array(g, dtype=np.int64).reshape((-1, k + 1)) for k, g in enumerate(groups)]
        self.SAPS = [OracleSAP(C[n], Ks[n], multi_groups[n], multi_costs[n]) for n in range(self.n_outputs)]
        self.sizes = [0] + [len(g) for g in self.groups]
        self.cumsizes = np.cumsum(self.sizes)
        self.L = int(self.cumsizes[-1])
        pos = {}
        for k, gk in enumerate(self.groups):
            for j, g in enumerate(gk):
                pos[tuple(int(x) for x in g)] = int(self.cumsizes[k]) + j
        self.mappings = [np.array([pos[tuple(int(x) for x in g)] for gk in self.SAPS[n].groups for g in gk],
                                  dtype=np.int64) for n in range(self.n_outputs)]
        allg = [g for gk in self.groups for g in gk]
        self.e = np.array([int(0 in g) for g in allg], dtype=np.int64)

    def variances(self, m, delta=0):
        """bluest/mosap.py:86-89"""
        return [self.SAPS[n].variance(m[self.mappings[n]], delta=delta) for n in range(self.n_outputs)]

    def variance_GH(self, m, nohess=False, delta=0):
        """bluest/mosap.py:91-100"""
        out = [self.SAPS[n].variance_GH(m[self.mappings[n]], nohess=nohess, delta=delta)
               for n in range(self.n_outputs)]
        return [o[0] for o in out], [o[1] for o in out], [o[2] for o in out]


# --------------------------------------------------------------------------------------------------
# bluest/spg.py:3-132, restated
# --------------------------------------------------------------------------------------------------

def linesearch(feval, x, f, g, d, last_fval, max_fevals, count):
    """bluest/spg.py:3-37: nonmonotone Armijo with safeguarded quadratic interpolation"""
    sigma_min, sigma_max, gamma = 0.1, 0.9, 1.0e-4
    fmax = max(last_fval)
    gdotd = g @ d
    alpha = 1.0
    xnew = x + alpha * d
    fnew = feval(xnew)
    count += 1
    while fnew > fmax + gamma * alpha * gdotd and count < max_fevals:
        if alpha <= sigma_min:
            alpha *= 0.5
        else:
            alpha_t = -0.5 * (alpha ** 2) * gdotd / (fnew - f - alpha * gdotd)
            if alpha_t < sigma_min or alpha_t > sigma_max * alpha:
                alpha_t = 0.5 * alpha
            alpha = alpha_t
        xnew = x + alpha * d
        fnew = feval(xnew)
        count += 1
    info = 0 if fnew <= fmax + gamma * alpha * gdotd else 2
    return count, fnew, xnew, info, alpha


def spg(feval, geval, proj, x, eps=1.0e-4, maxit=200, max_fevals=10 ** 5, lmbda_min=1e-30, lmbda_max=1e30,
        Hlength=10, trace=None):
    """bluest/spg.py:39-132.  `trace`, if a list, receives (it, f, gpmax, lmbda, alpha) per iteration."""
    it = 0
    count = 0
    last_fval = -np.inf * np.ones((Hlength,))
    x = proj(x)
    f = feval(x)
    g = geval(x)
    count += 1
    last_fval[0] = f
    gp = proj(x - g) - x
    gpmax = abs(gp).max()
    lmbda = min(lmbda_max, max(lmbda_min, 1.0 / gpmax)) if gpmax > 1.0e-15 else 0.0
    if trace is not None:
        trace.append((it, f, gpmax, lmbda, 0.0))
    while gpmax > eps and it < maxit and count < max_fevals:
        it += 1
        d = proj(x - lmbda * g) - x
        count, fnew, xnew, info, alpha = linesearch(feval, x, f, g, d, last_fval, max_fevals, count)
        if info == 2:
            return {"x": x, "f": f, "gpmax": gpmax, "it": it, "count": count, "solver_info": 2}
        f = fnew
        last_fval[it % Hlength] = f
        gnew = geval(xnew)
        s = xnew - x
        y = gnew - g
        sdots = s @ s
        sdoty = s @ y
        x = xnew
        g = gnew
        gp = proj(x - g) - x
        gpmax = abs(gp).max()
        lmbda = lmbda_max if sdoty <= 0 else min(lmbda_max, max(lmbda_min, sdots / sdoty))
        if trace is not None:
            trace.append((it, f, gpmax, lmbda, alpha))
    info = 0 if gpmax <= eps else (1 if it >= maxit else 2)
    return {"x": x, "f": f, "gpmax": gpmax, "it": it, "count": count, "solver_info": info}


def simplex_projection(v, z=1.0):
    """Euclidean projection onto {x >= 0, sum x = z} (sort-and-threshold; shift by max(v) first, the
    projection is shift-invariant and SPG feeds values of magnitude lmbda_max=1e30).  No reference
    counterpart: this is the build-defined `proj` for the new solver="spg"."""
    v = np.asarray(v, dtype=np.float64)
    u = v - v.max()
    s = np.sort(u)[::-1]
    css = np.cumsum(s) - z
    k = np.arange(1, len(s) + 1)
    cond = s - css / k > 0
    rho = k[cond][-1]
    tau = css[cond][-1] / rho
    return np.maximum(u - tau, 0.0)


def weighted_simplex_projection(u, s, z=1.0):
    """argmin sum (p-u)^2/s over {p >= 0, sum p = z}: p = max(u - tau*s, 0) (sort on the ratios u/s).
    Build-defined (scaled SPG step); with s = 1 it is simplex_projection."""
    u = np.asarray(u, dtype=np.float64)
    s = np.asarray(s, dtype=np.float64)
    r = u / s
    r = r - r.max()
    order = np.argsort(-r)
    rs, ss = r[order], s[order]
    c1 = np.cumsum(ss * rs)
    c0 = np.cumsum(ss)
    taus = (c1 - z) / c0
    k = np.nonzero(rs > taus)[0][-1]
    return s * np.maximum(r - taus[k], 0.0)
