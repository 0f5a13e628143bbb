"""
MOSAP -- multi-output sample-allocation problem on the GPU.  Mirrors bluest/mosap.py:18-123,291-344: same
constructor and attributes (SAPS, mappings, groups, sizes, cumsizes, L, e, ES), variances() / variance_GH()
with the reference's list-over-outputs return shape, solve() with solver="spg".

All outputs share ONE device plan: a call to variances()/variance_GH() is one Phi launch + one solve launch
(+ one gradient launch) for every output together, instead of the reference's serial loop over SAPS
(bluest/mosap.py:86-100).
"""
import weakref

import numpy as np
import torch

from . import misc
from .plan import EVAL_INF, EVAL_SINGULAR, Plan
from .host import phase_clock
from .sap import BLUESTError, LazyIndicators, SAP, SpgAllocator, restrict_plan, enforce_sample_caps, host_section, in_host_section, normalise_groups, sample_cap_rows, status_to_python


def _group_keys(gk, N):
    """one int64 key per group (mixed radix N); None if it could overflow"""
    k = gk.shape[1]
    if k == 0 or N ** k >= 2 ** 62:
        return None
    return gk @ (np.int64(N) ** np.arange(k, dtype=np.int64))


def build_mappings(groups, multi_groups, cumsizes, N=None):
    """mappings[n][j] = position in the global group list of the j-th group of output n (mosap.py:54-67).
    The reference searches the list linearly for every group (O(L_k^2)); here groups are keyed by one integer and
    located with a sorted search (vectorised), with a tuple dictionary as the fall-back for very large groups."""
    if N is None:
        N = 1 + max([int(g.max()) for g in groups if len(g)] + [0])
    key_cache = {}

    def sorted_keys(k):
        """(sorted keys, order) of the global groups of size k, built when an output really needs the search (outputs that use
        the global list itself take the identity map below: nothing is sorted at all in the usual case)"""
        if k not in key_cache:
            gk = groups[k - 1]
            keys = _group_keys(np.asarray(gk, dtype=np.int64).reshape(len(gk), -1), N) if len(gk) else None
            if keys is None:
                key_cache[k] = None
            else:
                order = np.argsort(keys, kind="stable")
                key_cache[k] = (keys[order], order)
        return key_cache[k]
    tuple_pos = None
    mappings = []
    for mg in multi_groups:
        idx = []
        for gk in mg:
            gk = np.asarray(gk, dtype=np.int64)
            if len(gk) == 0:
                continue
            k = gk.shape[1]
            base = int(cumsizes[k - 1])
            if k <= len(groups) and len(gk) == len(groups[k - 1]) and np.array_equal(gk, groups[k - 1]):
                idx.append(base + np.arange(len(gk), dtype=np.int64))          # identical list: identity map
                continue
            entry = sorted_keys(k) if k <= len(groups) else None
            keys = _group_keys(gk, N)
            if entry is not None and keys is not None:
                sk, order = entry
                pos = np.searchsorted(sk, keys)
                pos = np.minimum(pos, len(sk) - 1)
                if not (sk[pos] == keys).all():
                    missing = gk[np.nonzero(sk[pos] != keys)[0][0]]
                    raise AssertionError("group %s of an output is missing from the global group list (mosap.py:60)" % (tuple(missing),))
                idx.append(base + order[pos])
            else:
                if tuple_pos is None:
                    tuple_pos = {}
                    for kk, g_all in enumerate(groups):
                        for j, g in enumerate(np.asarray(g_all).tolist()):
                            tuple_pos[tuple(g)] = int(cumsizes[kk]) + j
                for g in gk.tolist():
                    if tuple(g) not in tuple_pos:
                        raise AssertionError("group %s of an output is missing from the global group list (mosap.py:60)" % (tuple(g),))
                    idx.append(np.array([tuple_pos[tuple(g)]], dtype=np.int64))
        mappings.append(np.concatenate(idx) if idx else np.zeros(0, dtype=np.int64))
    return mappings


class _LazyFlat(object):
    """flattened_groups (list of lists, mosap.py:31-37 / sap.py:66-83) without materialising ~K_tot Python lists: items are
    produced on demand (it is only read for the few groups of a reported allocation, blue_models.py:531)"""

    def __init__(self, groups):
        self._groups = groups
        self._cum = np.cumsum([0] + [len(g) for g in groups])

    def __getitem__(self, i):
        i = int(i)
        if i < 0:
            i += len(self)
        if not 0 <= i < len(self):
            raise IndexError(i)
        k = int(np.searchsorted(self._cum, i, side="right")) - 1
        return np.asarray(self._groups[k][i - int(self._cum[k])]).tolist()

    def __len__(self): return int(self._cum[-1])
    def __iter__(self): return (g for gk in self._groups for g in np.asarray(gk).tolist())
    def __eq__(self, other): return list(self) == list(other)
    def __repr__(self): return repr(list(self))


class _SapView(SAP):
    """SAPS[n] of a MOSAP: the reference builds one full SAP per output (mosap.py:39).  Here the per-output
    data already lives in the shared plan; this view gives the same attributes and closures, creating its own
    single-output plan only if somebody calls it directly."""

    def __init__(self, parent, n):
        self._parent_ref, self._n = weakref.ref(parent), n      # weak: MOSAP.SAPS -> view -> MOSAP would be a reference cycle
        self.verbose = parent.verbose
        self.C = parent.C[n]
        self.N = parent.N
        self.K = parent.Ks[n]
        self.costs = parent.multi_costs[n]
        self.samples = self.budget = self.eps = self.tot_cost = None
        groups = parent.multi_groups[n]
        self.sizes = [0] + [len(g) for g in groups]
        self.groups = groups
        self.flattened_groups = _LazyFlat(groups)
        self.cumsizes = np.cumsum(self.sizes)
        self.L = int(self.cumsizes[-1])
        self._plan_output = n
        self.ES = LazyIndicators(groups, self.N)
        self._psi = None
        self._plan = None

    @property
    def e(self):
        return self.ES[0]

    @property
    def _parent(self):
        parent = self._parent_ref()
        if parent is None:
            raise ReferenceError("the MOSAP this per-output view belongs to no longer exists")
        return parent

    @property
    def plan(self):
        if self._plan is None:
            self._plan = Plan(self.N, self.L, [{"K": self.K, "sizes": self.sizes[1:], "groups": self.groups,
                                                "invcovs": self.invcovs, "mapping": None}],
                              max_candidates=1, device=self._parent.plan.device)
        return self._plan

    @plan.setter
    def plan(self, value):
        self._plan = value

    def _inverse_source(self):
        return self._parent.plan.invcovs[self._n]

    def _inverse_plan(self):
        return self._parent.plan, self._n


class MOSAP(object):
    ''' MOSAP, MultiObjectiveSampleAllocationProblem '''

    def __init__(self, C, K, Ks, groups, multi_groups, costs, multi_costs, verbose=True, device=None, max_candidates=1):
        self.verbose = verbose
        self.n_outputs = len(C)
        self.C = C
        self.N = C[0].shape[0]
        self.K = K
        self.Ks = Ks
        self.costs = costs
        self.multi_groups = multi_groups
        self.multi_costs = multi_costs
        self.samples = None
        self.budget = None
        self.eps = None
        self.tot_cost = None
        clock = phase_clock("MOSAP.__init__")
        with host_section():
            clock.tick("host_section")
            self._build(C, K, Ks, groups, multi_groups, device, max_candidates, clock)
        clock.tick("exit")
        clock.report()
        self.setup_phases = clock.as_dict()     # where the constructor's wall-clock went (bench.py prints it for the cold set-up)

    def _build(self, C, K, Ks, groups, multi_groups, device, max_candidates, clock):
        normalise_groups(groups, K, flatten=False)                   # mosap.py:31-37 (in place)
        self.flattened_groups = _LazyFlat(groups)
        self.groups = groups
        for n in range(self.n_outputs):
            normalise_groups(multi_groups[n], Ks[n], flatten=False)  # sap.py:77 (in place, via SAP.__init__)

        clock.tick("normalise")
        self.sizes = [0] + [len(groupsk) for groupsk in groups]
        self.cumsizes = np.cumsum(self.sizes)
        self.L = int(self.cumsizes[-1])
        self.ES = LazyIndicators(groups, self.N)
        self.e = self.ES[0]
        clock.tick("indicators")
        self.mappings = build_mappings(groups, multi_groups, self.cumsizes, self.N)  # m[mappings[n]] = m_n

        outs = []
        self._identity_map = []                                   # output n lives on all groups in the global order
        flat_global = None                                        # ... and such outputs hand the plan ONE flattened copy of the global list
        for n in range(self.n_outputs):
            mg = multi_groups[n]
            ident = len(self.mappings[n]) == self.L and (self.mappings[n] == np.arange(self.L)).all()
            self._identity_map.append(bool(ident))
            gl = mg
            if ident and Ks[n] == K:
                if flat_global is None:
                    flat_global = np.ascontiguousarray(np.concatenate([np.asarray(g, dtype=np.int64).ravel() for g in groups]))
                gl = flat_global
            outs.append({"K": Ks[n], "sizes": [len(g) for g in mg], "groups": gl, "C": np.asarray(C[n], dtype=np.float64),
                         "mapping": None if ident else self.mappings[n]})
        clock.tick("mappings")
        self.plan = Plan(self.N, self.L, outs, max_candidates=max_candidates, device=device)
        clock.tick("Plan")
        self.SAPS = [_SapView(self, n) for n in range(self.n_outputs)]
        clock.tick("views")

    def check_input(self, budget, eps):
        """bluest/mosap.py:74-84"""
        if budget is None and eps is None:
            raise ValueError("Need to specify either budget or RMSE tolerance")
        if eps is not None:
            try:
                if len(eps) != self.n_outputs:
                    raise ValueError("eps must be a scalar or an array of tolerances")
                eps = np.array(eps)
            except TypeError:
                eps = np.array([eps for n in range(self.n_outputs)])
        return budget, eps

    def variances(self, m, delta=0):
        """bluest/mosap.py:86-89: list over outputs of V_n(m[mappings[n]])"""
        var, _, status = self.plan.eval(m, delta=delta, want_grad=False)
        var = var[0].cpu().numpy()
        status = status[0].cpu().numpy()
        out = []
        for n in range(self.n_outputs):
            if status[n] == EVAL_INF:
                out.append(np.inf)
                continue
            status_to_python(int(status[n]), "variances[output %d]" % n)
            out.append(float(var[n]))
        return out

    def variance_GH(self, m, nohess=False, delta=0):
        """bluest/mosap.py:91-100: (variances, gradients, hessians) as lists over outputs; gradient n is with
        respect to m[mappings[n]]"""
        host = not isinstance(m, torch.Tensor)
        var, grad, status = self.plan.eval(m, delta=delta, want_grad=True)
        var = var[0].cpu().numpy()
        status = status[0].cpu().numpy()
        g_all = grad[0].cpu().numpy() if host else grad[0]
        if (status == EVAL_SINGULAR).any():
            # rank-deficient restricted Phi of some output: the reference's pinv semantics (misc.py:487,490) for those outputs
            var_p, grad_p, status_p = self.plan.eval_pinv(m, delta=delta)
            var_p, status_p = var_p[0].cpu().numpy(), status_p[0].cpu().numpy()
            g_p = grad_p[0].cpu().numpy() if host else grad_p[0]
            g_all = g_all.copy() if host else g_all.clone()
            for n in np.flatnonzero(status == EVAL_SINGULAR):
                off, Ln = self.plan.grad_off[n], len(self.mappings[n])
                g_all[off:off + Ln] = g_p[off:off + Ln]
                var[n], status[n] = var_p[n], status_p[n]
        variances, gradients, hessians = [], [], []
        for n in range(self.n_outputs):
            off, Ln = self.plan.grad_off[n], len(self.mappings[n])
            gradients.append(g_all[off:off + Ln])
            if status[n] == EVAL_INF:
                variances.append(np.inf)
                hessians.append(None)
                continue
            variances.append(float(var[n]))
            if nohess:
                hessians.append(None)
            else:
                m_h = m.cpu().numpy() if isinstance(m, torch.Tensor) else np.asarray(m, dtype=np.float64)
                hessians.append(self.SAPS[n]._hessian(m_h[self.mappings[n]], delta))
        return variances, gradients, hessians

    def _restricted_plan(self, keep):
        """shared plan of this problem restricted to the global groups `keep` (sorted indices): built natively from the shared
        plan (bluest_plan_restrict: group lists filtered on the host, stored pseudo-inverses gathered on the device)"""
        return restrict_plan(self.plan, np.asarray(keep, dtype=np.int64))

    def get_cleanup_matrices(self, m, delta=0, columns=None):
        """bluest/mosap.py:102-111: the per-output cleanup matrices stacked, (n_outputs*N, L).  Phi of every output comes
        from ONE launch of the shared plan; the per-group products run through the `cleanupK` kernel.
        columns (extension): global group indices -- only those columns are computed and returned, (n_outputs*N, len(columns));
        `cleanup_solution` only ever looks at the support of m."""
        m_h = np.asarray(m.cpu().numpy() if isinstance(m, torch.Tensor) else m, dtype=np.float64)
        PHI = self.plan.phi_matrix(m_h, delta=delta)[0].cpu().numpy()
        cols = None if columns is None else np.asarray(columns, dtype=np.int64)
        Xs = []
        for n in range(self.n_outputs):
            sap = self.SAPS[n]
            if abs(m_h[self.mappings[n]]).max() < 0.05:
                raise ValueError("No entry greater or equal than 1 found in m.")
            invPHI = np.linalg.pinv(PHI[n])
            if cols is None:
                X = np.zeros((self.N, self.L))
                X[:, self.mappings[n]] = np.hstack([misc.cleanupK(k, sap.sizes[k], sap.groups[k - 1], sap.invcovs[k - 1], invPHI)
                                                    for k in range(1, sap.K + 1) if sap.sizes[k] > 0])
            else:
                X = np.zeros((self.N, len(cols)))
                local = np.full(self.L, -1, dtype=np.int64)
                local[self.mappings[n]] = np.arange(len(self.mappings[n]))
                loc = local[cols]                                   # position of each requested group inside output n (-1: absent)
                for k in range(1, sap.K + 1):
                    lo, hi = sap.cumsizes[k - 1], sap.cumsizes[k]
                    pick = np.flatnonzero((loc >= lo) & (loc < hi))
                    if len(pick) == 0:
                        continue
                    sel = loc[pick] - lo
                    gk = np.asarray(sap.groups[k - 1])[sel]
                    ick = np.asarray(sap.invcovs[k - 1]).reshape(-1, k * k)[sel].ravel()
                    X[:, pick] = misc.cleanupK(k, len(sel), gk, ick, invPHI)
            Xs.append(X)
        return np.vstack(Xs)

    def cleanup_solution(self, m, delta=0, tol=0):
        """bluest/mosap.py:125-210: look for a sparser allocation with the same (max) variance and no larger cost.  Moves
        along null vectors of the cleanup matrix restricted to the support, cheapest direction first, as far as
        non-negativity and "model 0 of every output is still sampled once" allow; a move is kept if the variance does not
        get worse by more than 1e-4 relative.  Every variance is one evaluation of the shared plan on the GPU; the
        null-space is scipy's (as in the reference)."""
        from scipy.linalg import null_space
        m = np.array(m, dtype=np.float64)
        N, L, w = self.N, self.L, self.costs
        E = np.zeros((self.n_outputs, L))
        for n in range(self.n_outputs):
            E[n, self.mappings[n]] = self.e[self.mappings[n]]

        def report(it, nnz, nullsize, V):
            if self.verbose:
                ns = "n/a" if nullsize < 0 else "%3d" % nullsize
                print("It %3d: Solution cleanup, L = %d, N = %d, nnz = %d, nullspace size = %s, variance = %e." % (it, L, N, nnz, ns, V))

        support = np.flatnonzero(m > tol)
        V0 = V = max(self.variances(m, delta=delta))
        it, nullsize = 0, -1
        if self.verbose: print("\nSolution cleanup started!")
        report(it, len(support), nullsize, V)
        while len(support) > N:
            support = np.flatnonzero(m > tol)
            m[m < tol] = 0
            if it > 0 and L >= 1000: report(it, len(support), nullsize, V)
            it += 1
            directions = null_space(self.get_cleanup_matrices(m, delta=delta, columns=support))
            dcost = w[support] @ directions
            directions = directions[:, dcost != 0] * -np.sign(dcost[dcost != 0])      # every direction now lowers the cost
            dcost = -abs(dcost[dcost != 0])
            nullsize = len(dcost)
            if nullsize == 0:
                break
            Er = E[:, support]
            sampled0 = Er @ m[support]
            step = 0.0
            for j in np.argsort(abs(dcost))[::-1]:
                t = directions[:, j]
                Et = Er @ t
                lim0 = np.min(abs(sampled0[Et < 0] - 1) / abs(Et[Et < 0])) if (Et < 0).any() else np.inf
                limp = np.min(m[support][t < 0] / abs(t[t < 0])) if (t < 0).any() else np.inf
                step = max(min(lim0, limp), 0)
                if step > 5 * tol:
                    trial = m.copy()
                    trial[support] += step * t
                    V = max(self.variances(trial, delta=delta))
                    if V < V0 or abs(V - V0) / abs(V0) < 1.0e-4:
                        m = trial
                        break
                    step = 0.0
            if step <= 5 * tol:
                break
            V = max(self.variances(m, delta=delta))
        support = np.flatnonzero(m > tol)
        m[m < tol] = 0
        V = max(self.variances(m, delta=delta))
        report(it, len(support), nullsize, V)
        if self.verbose: print("Solution cleanup completed.\n")
        return m

    def compute_BLUE_estimators(self, sums, samples):
        """bluest/mosap.py:113-123"""
        out = []
        for n in range(self.n_outputs):
            sums_n = [sums[n][item] for item in self.mappings[n]]
            out.append(self.SAPS[n].compute_BLUE_estimator(sums_n, samples=np.asarray(samples)[self.mappings[n]]))
        mus = [item[0] for item in out]
        Vars = np.array([item[1] for item in out])
        return mus, Vars

    def get_max_sample_constraints(self, max_model_samples):
        """bluest/mosap.py:326-344"""
        return sample_cap_rows(self.ES, self.N, max_model_samples)

    @in_host_section
    def solve(self, budget=None, eps=None, solver="spg", x0=None, continuous_relaxation=False, max_model_samples=None,
              solver_params=None):
        """bluest/mosap.py:291-324 with solver="spg" """
        if budget is None and eps is None:
            raise ValueError("Need to specify either budget or RMSE tolerance")
        if solver not in ["spg", "scipy", "cvxpy", "ipopt", "cvxopt"]:
            raise ValueError("Optimization solvers available: 'spg' (this build); the reference also lists 'scipy', 'ipopt', 'cvxopt', 'cvxpy'")
        if solver != "spg":
            raise BLUESTError("solver=%r is a third-party back-end of the reference that this GPU build does not ship; use solver='spg'" % solver)
        cap_rows, cap_rhs = self.get_max_sample_constraints(max_model_samples)   # validates the argument as the reference does
        budget, eps = self.check_input(budget, eps)

        if self.verbose:
            if eps is None: print("Minimizing statistical error for fixed cost...\n")
            else:           print("Minimizing cost given statistical error tolerance...\n")

        es, e_all = [], None
        for n in range(self.n_outputs):
            if self._identity_map[n]:                             # (the usual case: one vector for all outputs, no gather / scatter)
                if e_all is None:
                    e_all = np.asarray(self.e, dtype=np.float64)
                es.append(e_all)
                continue
            ee = np.zeros((self.L,))
            ee[self.mappings[n]] = self.e[self.mappings[n]]
            es.append(ee)
        alloc = SpgAllocator(self.plan, self.costs, es, verbose=False, subplan=self._restricted_plan)
        try:
            samples = alloc.solve(budget=budget, eps=eps, x0=x0, params=solver_params)
            self.solver_info = alloc.info
            cap_models = None if max_model_samples is None else [i for i in range(self.N) if np.isfinite(max_model_samples[i])]
            samples = enforce_sample_caps(self.plan, self.costs, cap_rows, cap_rhs, samples, budget, eps, solver_params, self,
                                          cap_models=cap_models)
        except BLUESTError as err:
            if self.verbose: print(str(err))
            self.samples = None
            return None
        if any(samples @ ee < 1.0 - 1.0e-9 for ee in {id(ee): ee for ee in es}.values()):
            if self.verbose: print("SPG solution samples model 0 less than once for some output; infeasible.")
            self.samples = None
            return None

        if not continuous_relaxation:
            from .integer import integer_projection_mosap
            try:
                samples = integer_projection_mosap(self, samples, budget=budget, eps=eps, max_model_samples=max_model_samples)
            except AssertionError as err:
                print(str(err))
                self.samples = None
                return None

        self.samples = samples
        self.budget = budget
        self.eps = eps
        self.tot_cost = samples @ self.costs
        for n in range(self.n_outputs):
            self.SAPS[n].samples = samples.copy() if self._identity_map[n] else samples[self.mappings[n]]
        return samples
